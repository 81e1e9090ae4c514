// see cv_image_shim.h: stand-in used only for the oracle/_ref build of the reference sources
#pragma once
#include "opencv2/core/core.hpp"
#include "cv_image_shim.h"
