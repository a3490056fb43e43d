// Internal helpers shared by the translation units of libbasic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/basic_hip.h"

namespace basic {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define BASIC_HIP_TRY(expr)                                                      \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) return ::basic::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define BASIC_REQUIRE(cond, msg)            \
    do {                                    \
        if (!(cond)) {                      \
            ::basic::set_error(msg);        \
            return BASIC_ERR_INVALID;       \
        }                                   \
    } while (0)

// Fails (BASIC_ERR_NO_DEVICE) unless a HIP device is usable.  Never falls back to the CPU.
int require_device();

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Raises a kernel's dynamic-LDS limit to the CU's 160 KiB, once per (kernel, device): the attribute belongs to the
// device's copy of the code object, so a second GPU used by the same process needs its own call.  Thread-safe.
hipError_t ensure_max_lds(const void *kernel);

// Wavefronts (streams) per workgroup of the calling thread's next batched fast rANS launches (0 = default); returns the
// previous setting.
int set_rans_waves(int waves_per_block);

// Whether the calling thread's next launches of the persistent first-layer convolution hand their tiles out dynamically (a
// device counter) instead of by a static stride; returns the previous setting.  Dynamic pays when other HIP streams hold
// compute units for long (a late workgroup no longer drags the launch), static is faster alone on the chip.
bool set_dynamic_tiles(bool on);

constexpr int kWave = 64;  // CDNA wavefront

// Canonical summation block of the masked convolution (mconv.hip header comment): channels of one (tap, input group) slab
// whose products form ONE fp32 MFMA / FMA chain; the persistent scan-line kernel (scanline.hip) sums in the same blocks.
#define BASIC_MCONV_BLOCK_CHANNELS 64

// Device view of a table set's fast-decoder search image (rans.hip), for kernels outside rans.hip that decode in place.
struct RansFastView {
    const uint32_t *image = nullptr, *meta = nullptr;   // image: per row 64 x {key, start, freq, pad} (rows <= 64 entries)
    const int32_t *sizes = nullptr, *offsets = nullptr; //        or {dummy lane, 64 block-end probes, the row}; meta: byte offsets
    int image_words = 0, rows = 0, precision = 16, bypass = 1, bypass_precision = 4;
};
int rans_fast_view(const basic_rans_tables *t, RansFastView *out);   // BASIC_ERR_INVALID when the set has no fast image

}  // namespace basic
