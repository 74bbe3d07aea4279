#pragma once
#include "opencv2/core/core.hpp"
