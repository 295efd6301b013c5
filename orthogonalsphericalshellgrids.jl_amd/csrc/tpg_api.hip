// tpg_api.hip -- version / error channel / argument validation of libtripolar_hip.
#include "tpg_common.hpp"
#include <string.h>

namespace tpg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_geom(int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft)
{
    if (ft != TPG_F32 && ft != TPG_F64) { set_error("unknown element type ft=%d", ft); return TPG_ERR_INVALID_ARGUMENT; }
    if (Nx < 2 || Ny < 1 || Nz < 1 || Hx < 0 || Hy < 0 || Hz < 0) {
        set_error("invalid size (%d,%d,%d) / halo (%d,%d,%d)", Nx, Ny, Nz, Hx, Hy, Hz);
        return TPG_ERR_INVALID_ARGUMENT;
    }
    if (Nx % 2) {   // src/tripolar_grid.jl:81-83
        set_error("The number of cells in the longitude dimension should be even!");
        return TPG_ERR_ODD_NLAMBDA;
    }
    if (Hx > Nx || Hy > Ny) {   // Oceananigans requires halo <= size in every non-Flat direction
        set_error("halo (%d,%d) larger than size (%d,%d)", Hx, Hy, Nx, Ny);
        return TPG_ERR_UNSUPPORTED;
    }
    // 32-bit work-item indices inside one (field, level) slab; 64-bit element offsets everywhere
    if ((long long)(Nx + 2 * Hx) * (Ny + 2 * Hy) >= (1ll << 31) || (long long)Nx * (Nz + 2 * Hz) * (Hy + 1) >= (1ll << 31)) {
        set_error("horizontal plane too large for 32-bit indexing");
        return TPG_ERR_UNSUPPORTED;
    }
    return TPG_OK;
}

}  // namespace tpg

extern "C" {

int tpg_version(void) { return TPG_VERSION; }

const char* tpg_last_error(void) { return tpg::g_err; }

const char* tpg_status_string(int status)
{
    switch (status) {
    case TPG_OK: return "ok";
    case TPG_ERR_INVALID_ARGUMENT: return "invalid argument";
    case TPG_ERR_ODD_NLAMBDA: return "The number of cells in the longitude dimension should be even!";
    case TPG_ERR_BAD_PARTITION: return "latitude band outside the global grid (only y-partitioning is supported)";
    case TPG_ERR_WORKSPACE: return "workspace missing or too small";
    case TPG_ERR_UNSUPPORTED: return "unsupported size";
    case TPG_ERR_NOT_NORTH: return "Zipper boundary condition is valid on the north side only";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
    }
}

}  // extern "C"
