// pnr_common.h -- error plumbing shared by the translation units of libpixelnerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pixelnerf_hip.h"

// records msg for pnr_last_error() and returns code
int pnr_fail(int code, const char *msg);
// hipGetLastError() after a launch -> PNR_OK / PNR_E_HIP (with message)
int pnr_check_launch(const char *where);
int pnr_check_hip(hipError_t e, const char *where);
