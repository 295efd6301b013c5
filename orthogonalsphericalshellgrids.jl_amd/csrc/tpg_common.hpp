// tpg_common.hpp -- shared host-side helpers of libtripolar_hip (error channel, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/tripolar_hip.h"

namespace tpg {

void set_error(const char* fmt, ...);

// returns a C-ABI status from a HIP error (positive hipError_t) and records the message
inline int hip_status(hipError_t e, const char* what)
{
    if (e == hipSuccess) return TPG_OK;
    set_error("%s: %s (hipError_t %d)", what, hipGetErrorString(e), (int)e);
    return (int)e;
}

inline int launch_status(const char* kernel)
{
    return hip_status(hipGetLastError(), kernel);
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// geometry of one padded 3-D field (or 2-D with Nz = 1, Hz = 0)
struct Geom {
    int Nx, Ny, Nz, Hx, Hy, Hz;
    int sx;          // Nx + 2Hx
    int sy;          // Ny + 2Hy
    long long plane; // sx * sy
};

inline Geom make_geom(int Nx, int Ny, int Nz, int Hx, int Hy, int Hz)
{
    Geom g{ Nx, Ny, Nz, Hx, Hy, Hz, Nx + 2 * Hx, Ny + 2 * Hy, 0 };
    g.plane = (long long)g.sx * g.sy;
    return g;
}

int check_geom(int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft);

// Kernel-selection record.  The product library (libtripolar_hip.so) carries ONE constant record -- the defaults below, no
// environment access.  The test library (libtripolar_hip_test.so, -DTPG_TEST_ABI) reads the TPG_* cross-check knobs of
// include/tripolar_hip_test.h from the environment once into an immutable record; tpg_reload_config() publishes a fresh one.
struct Config {
    int cells_variant;        // TPG_CELLS_VARIANT  2 LDS-tile kernel + k_halos (default; the product's only form), 3 LDS-tile kernel that also writes the
                              //                    halo cells (k_cells_tile_push, test library only), 0 thread-per-cell cross-check
    bool build_nt;            // TPG_BUILD_NT       1 streaming stores in tpg_build_grid (default), 0 plain
    int zipper_variant;       // TPG_ZIPPER_VARIANT 3 column items (default), 0 row items (the fallback kernels)
    int fill_fused;           // TPG_FILL_FUSED     -1 automatic (default), 0 never, 1 wherever valid, 2 wherever valid in the one-thread-per-cell form
    int fill_merged;          // TPG_FILL_MERGED    -1 automatic (default), 0 never, 1 wherever valid
    bool exchange_in_capture; // TPG_EXCHANGE_IN_CAPTURE  0 (default): the RCCL seam exchange refuses a capturing stream; 1: lets it through
                              //                          (tools/rccl_capture_probe.py, the diagnostic of the round-2 capture stall)
    int exchange_fail_stage;  // TPG_EXCHANGE_FAIL_STAGE  -1 (default, and always in the product): off; k >= 0: the pipelined seam exchange reports
                              //                          an injected failure right after the RCCL group of stage k went onto comm_stream
                              //                          (tests/test_gpu_exchange.py: the error-path post-condition)
    const char* rccl_library; // TPG_RCCL_LIBRARY  nullptr (default, and always in the product): bind librccl by its fixed names; a path: bind THAT
                              //                   library instead -- the test double tools/nccl_shim/libnccl_shim.so, so that the exchange code runs
                              //                   between several real processes on a one-GPU box (RCCL refuses two ranks on one device)
};
const Config& config();

}  // namespace tpg
