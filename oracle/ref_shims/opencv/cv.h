#pragma once
#include "../opencv2/imgproc/imgproc.hpp"
