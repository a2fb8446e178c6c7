// pnr_api.hip -- error plumbing and library facts for libpixelnerf_hip.so.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pnr_common.h"
#include "pnr_internal.h"
#include "pnr_layout.h"

static thread_local char g_err[512] = "";

int pnr_fail(int code, const char *msg) {
    std::snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

int pnr_check_hip(hipError_t e, const char *where) {
    if (e == hipSuccess) return PNR_OK;
    std::snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return PNR_E_HIP;
}

int pnr_check_launch(const char *where) { return pnr_check_hip(hipGetLastError(), where); }

extern "C" const char *pnr_last_error(void) { return g_err; }

extern "C" int pnr_version(int *major, int *minor) {
    if (major) *major = 0;
    if (minor) *minor = 1;
    return PNR_OK;
}

// A library built with experiment switches (tools/build_variant.sh passes -DPNR_VARIANT next to them; the kernel sources only
// honour a switch under `#if defined(PNR_VARIANT) && defined(...)`) identifies itself with a NEGATIVE revision: the product
// binding refuses it (pixelnerf_amd/_lib.py), the A/B tools opt in.
extern "C" int pnr_abi_version(void) {
#ifdef PNR_VARIANT
    return -PNR_ABI_VERSION;
#else
    return PNR_ABI_VERSION;
#endif
}

extern "C" int pnr_device_info(int *num_cus, int *lds_bytes_per_block) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return pnr_check_hip(e, "hipGetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return pnr_check_hip(e, "hipGetDeviceProperties");
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (lds_bytes_per_block) *lds_bytes_per_block = pnr::LDS_TOTAL;
    return PNR_OK;
}

int pnr::device_xcd_count() {
    // the override is read per call (a getenv, no device work): tests switch the tile order inside one process
    if (const char *ov = std::getenv("PIXELNERF_XCD_COUNT")) {
        const int n = std::atoi(ov);
        return n > 0 ? n : 0;
    }
    static int cached[64];  // per device ordinal; 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || n <= 0) n = 1;
        cached[dev] = n;
    }
    return cached[dev];
}

// ---- fp16-range guard of the fp32-class ("f16x3") kernels
static thread_local unsigned int *g_sat_flags = nullptr;
static thread_local int g_sat_slot = 0;

unsigned int *pnr::saturation_guard_word() { return g_sat_flags ? g_sat_flags + g_sat_slot : nullptr; }
void pnr::saturation_guard_slot(int slot) { g_sat_slot = slot == 1 ? 1 : 0; }

extern "C" int pnr_saturation_guard(unsigned int *flags_dev) {
    g_sat_flags = flags_dev;
    g_sat_slot = 0;
    return PNR_OK;
}
