// forwards to the one stand-in header of this directory
#pragma once
#include "gslam_mock_api.h"
