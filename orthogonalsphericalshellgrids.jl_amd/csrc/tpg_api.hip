// tpg_api.hip -- version / error channel / argument validation of libtripolar_hip.
#include "tpg_common.hpp"
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

namespace tpg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifdef TPG_TEST_ABI
// libtripolar_hip_test.so only: the cross-check / tuning knobs are read from the environment once, at the first call into
// the library, into an immutable record; tpg_reload_config() publishes a fresh record atomically (tests).
static std::atomic<const Config*> g_config{nullptr};
static std::mutex g_config_mutex;

static int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

static const Config* read_config()
{
    Config* c = new Config;         // a few bytes per (re)load, never freed: readers may still hold the old record
    { const int v = env_int("TPG_CELLS_VARIANT", 2); c->cells_variant = (v == 0 || v == 3) ? v : 2; }
    c->build_nt = env_int("TPG_BUILD_NT", 1) != 0;
    c->zipper_variant = env_int("TPG_ZIPPER_VARIANT", 3) == 0 ? 0 : 3;
    c->fill_fused = env_int("TPG_FILL_FUSED", -1);
    c->fill_merged = env_int("TPG_FILL_MERGED", -1);
    c->exchange_in_capture = env_int("TPG_EXCHANGE_IN_CAPTURE", 0) != 0;
    c->exchange_fail_stage = env_int("TPG_EXCHANGE_FAIL_STAGE", -1);
    const char* lib = getenv("TPG_RCCL_LIBRARY");
    c->rccl_library = (lib && *lib) ? strdup(lib) : nullptr;
    return c;
}

const Config& config()
{
    const Config* c = g_config.load(std::memory_order_acquire);
    if (!c) {
        std::lock_guard<std::mutex> lock(g_config_mutex);
        c = g_config.load(std::memory_order_acquire);
        if (!c) { c = read_config(); g_config.store(c, std::memory_order_release); }
    }
    return *c;
}
#else
// the product library has no knobs: one constant record, no environment access
const Config& config()
{
    static const Config k{ 2, true, 3, -1, -1, false, -1, nullptr };
    return k;
}
#endif

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

#ifdef TPG_TEST_ABI
int tpg_reload_config(void)
{
    std::lock_guard<std::mutex> lock(tpg::g_config_mutex);
    tpg::g_config.store(tpg::read_config(), std::memory_order_release);
    return TPG_OK;
}
#endif

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
    case TPG_ERR_RCCL: return "RCCL error (librccl missing or ncclResult_t failure)";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
    }
}

}  // extern "C"
