#pragma once
#include "../../../g2o_standin.hpp"
