#pragma once
#include "imgproc/imgproc.hpp"
