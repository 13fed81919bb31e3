#pragma once
#include "../../../ref_shims/opencv2/core/core.hpp"
