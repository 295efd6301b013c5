// tpg_zipper.hip -- north-seam Zipper halo fill (index reversal + vector sign flip) for gfx950,
// plus the periodic-x pass and the latitude-band pack/unpack kernels that sit either side of it.
//
// Replaces fold_north_{center_center,face_center,center_face,face_face}!
// (src/zipper_boundary_condition.jl:70-138) as invoked per (i,k) by _fill_north_halo! (:146-155).
//
// HBM-bound integer/index work, no MFMA.  Design:
//  * one launch folds a whole BATCH of fields (pointer table in the kernarg segment), all levels;
//  * a work item is one 16-byte chunk (2 doubles / 4 floats) of one destination row: rows
//    Ny+1..Ny+Hy, plus the upper half of row Ny for y-Center fields (the row-Ny substitution);
//  * consecutive lanes own consecutive destination chunks (ascending, 16-B aligned stores) and
//    read the mirrored source chunk (descending addresses, one contiguous 1 KiB window per wave),
//    reverse it in registers and multiply by the sign;
//  * x-Face rows mirror about an odd offset (i' = Nx - i + 2): their source windows are 8-B (f64)
//    / 4-B (f32) off 16-B alignment, read with dword-aligned wide loads;
//  * a 1-D grid of 256-thread blocks, grid-stride free (one item per thread), 64-bit element
//    offsets, 32-bit item indices.
// Geometries whose rows do not split into 16-B aligned chunks (odd Hx -- the reference's model halo (5, 5, 5) --, Float32 with Nx = 2 mod 4,
// 16-B-misaligned base pointers) run the same kernels in their GEN form: the same chunks, stored element-aligned (tpg_zipper_kernels.hpp).
#include "tpg_zipper_kernels.hpp"

namespace tpg {
thread_local hipEvent_t ev_start = nullptr, ev_stop = nullptr;
}


extern "C" {

int tpg_zipper_fill(void* const fields[], int nfields, const int8_t xloc[], const int8_t yloc[],
                    const int32_t sign[], int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                    int kstart, int kcount, int ft, void* stream)
{
    int rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft);
    if (rc) return rc;
    if ((rc = check_fields(fields, nfields))) return rc;
    if (!xloc || !yloc || !sign) { tpg::set_error("null location/sign table"); return TPG_ERR_INVALID_ARGUMENT; }
    for (int f = 0; f < nfields; ++f)
        if ((xloc[f] != TPG_CENTER && xloc[f] != TPG_FACE) || (yloc[f] != TPG_CENTER && yloc[f] != TPG_FACE)) {
            // _fill_north_halo! has methods for the four (x,y) location pairs only (:140-155)
            tpg::set_error("field %d: no zipper method for location (%d,%d)", f, xloc[f], yloc[f]);
            return TPG_ERR_INVALID_ARGUMENT;
        }
    if (kcount < 0 || kstart < 1 - Hz || kstart + kcount - 1 > Nz + Hz) {
        tpg::set_error("level range %d:%d outside %d:%d", kstart, kstart + kcount - 1, 1 - Hz, Nz + Hz);
        return TPG_ERR_INVALID_ARGUMENT;
    }
    // Hy = 0 leaves no halo rows to fold, but the row-Ny substitution of the y-Center folds is outside the
    // j loop of the reference (zipper_boundary_condition.jl:102,135) and still applies
    if (kcount == 0) return TPG_OK;
    Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
    hipStream_t s = tpg::as_stream(stream);
    for (int f0 = 0; f0 < nfields; f0 += TPG_MAX_FIELDS) {
        int n = nfields - f0 < TPG_MAX_FIELDS ? nfields - f0 : TPG_MAX_FIELDS;
        rc = (ft == TPG_F64) ? zipper_batch<double>(fields + f0, n, xloc + f0, yloc + f0, sign + f0, g, kstart, kcount, s)
                             : zipper_batch<float>(fields + f0, n, xloc + f0, yloc + f0, sign + f0, g, kstart, kcount, s);
        if (rc) return rc;
    }
    return TPG_OK;
}

int tpg_zipper_fill_timed(void* const fields[], int nfields, const int8_t xloc[], const int8_t yloc[],
                          const int32_t sign[], int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                          int kstart, int kcount, int ft, void* stream, void* start_event, void* stop_event)
{
    if (nfields > TPG_MAX_FIELDS) { tpg::set_error("timed launch: at most %d fields (one kernel)", TPG_MAX_FIELDS); return TPG_ERR_UNSUPPORTED; }
    tpg::ev_start = static_cast<hipEvent_t>(start_event);
    tpg::ev_stop = static_cast<hipEvent_t>(stop_event);
    int rc = tpg_zipper_fill(fields, nfields, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, kstart, kcount, ft, stream);
    tpg::ev_start = tpg::ev_stop = nullptr;
    return rc;
}

int tpg_event_create(void** event)
{
    if (!event) { tpg::set_error("null event pointer"); return TPG_ERR_INVALID_ARGUMENT; }
    hipEvent_t e;
    // timing-only events: no system-scope fence when they complete (a default event attached to a launch turns the
    // kernel's end-of-dispatch release into a system-scope one, which lands inside the measured interval)
    int rc = tpg::hip_status(hipEventCreateWithFlags(&e, hipEventDisableSystemFence), "hipEventCreateWithFlags");
    *event = rc ? nullptr : e;
    return rc;
}

int tpg_event_destroy(void* event)
{
    return event ? tpg::hip_status(hipEventDestroy(static_cast<hipEvent_t>(event)), "hipEventDestroy") : TPG_OK;
}

int tpg_event_elapsed_ms(void* start_event, void* stop_event, float* ms)
{
    if (!start_event || !stop_event || !ms) { tpg::set_error("null event"); return TPG_ERR_INVALID_ARGUMENT; }
    int rc = tpg::hip_status(hipEventSynchronize(static_cast<hipEvent_t>(stop_event)), "hipEventSynchronize");
    if (rc) return rc;
    return tpg::hip_status(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)), "hipEventElapsedTime");
}

int tpg_periodic_x_fill(void* const fields[], int nfields, int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                        int ft, void* stream)
{
    int rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft);
    if (rc) return rc;
    if ((rc = check_fields(fields, nfields))) return rc;
    if (Hx == 0) return TPG_OK;
    Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
    hipStream_t s = tpg::as_stream(stream);
    for (int f0 = 0; f0 < nfields; f0 += TPG_MAX_FIELDS) {
        int n = nfields - f0 < TPG_MAX_FIELDS ? nfields - f0 : TPG_MAX_FIELDS;
        PtrTable pt;
        for (int f = 0; f < n; ++f) pt.ptr[f] = fields[f0 + f];
        PerArgs a{ Nx, Hx, g.sx, (long long)g.sy * (Nz + 2 * Hz), n };
        const int epc = ft == TPG_F64 ? 2 : 4;
        bool vec = (Hx % epc == 0) && (Nx % epc == 0);
        for (int f = 0; f < n && vec; ++f) vec = ((uintptr_t)pt.ptr[f] % 16) == 0;
        if (vec) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const int cpr = Hx / epc;
            dim3 gridv((unsigned)((a.nrows * cpr + 255) / 256), (unsigned)n);
            TPG_LAUNCH(k_periodic_x_vec<u32x4>, gridv, dim3(256), s, pt, a, cpr, epc);
        } else {
            long long total = a.nrows * Hx * n;
            dim3 grid((unsigned)((total + 255) / 256));
            if (ft == TPG_F64) TPG_LAUNCH(k_periodic_x<double>, grid, dim3(256), s, pt, a);
            else               TPG_LAUNCH(k_periodic_x<float>, grid, dim3(256), s, pt, a);
        }
        if ((rc = tpg::launch_status("k_periodic_x"))) return rc;
    }
    return TPG_OK;
}

int tpg_fill_halo_regions(void* const fields[], int nfields, const int8_t xloc[], const int8_t yloc[],
                          const int32_t sign[], int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                          int north_is_zipper, int ft, void* stream)
{
    int rc = TPG_OK;
    // small fields: one fused launch (k_fill_fused); TPG_FILL_FUSED=0 never, =1 whenever the geometry allows
    const int mode = tpg::config().fill_fused;
    if (north_is_zipper && mode != 0 && Hx > 0 && Hy > 0 && Nx >= 2 * Hx + 2 && Ny >= 2 * Hy + 2) {
        const long long per_level = (long long)(Hy + 1) * (Nx + 2 * Hx) + 2ll * Hx * (Ny + Hy - 1);
        const long long items = per_level * (Nz + 2 * Hz);
        if (items < (1ll << 31) && (mode >= 1 || items * nfields <= (1ll << 20))) {
            if ((rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft))) return rc;
            if ((rc = check_fields(fields, nfields))) return rc;
            if (!xloc || !yloc || !sign) { tpg::set_error("null location/sign table"); return TPG_ERR_INVALID_ARGUMENT; }
            for (int f = 0; f < nfields; ++f)
                if ((xloc[f] != TPG_CENTER && xloc[f] != TPG_FACE) || (yloc[f] != TPG_CENTER && yloc[f] != TPG_FACE)) {
                    tpg::set_error("field %d: no zipper method for location (%d,%d)", f, xloc[f], yloc[f]);
                    return TPG_ERR_INVALID_ARGUMENT;
                }
            Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
            FusedArgs a{ Nx, Ny, Hx, Hy, Hz, Nz, g.sx, g.sy, (long long)g.sx * g.sy, (int)per_level };
            hipStream_t s = tpg::as_stream(stream);
            for (int f0 = 0; f0 < nfields; f0 += TPG_MAX_FIELDS) {
                const int n = nfields - f0 < TPG_MAX_FIELDS ? nfields - f0 : TPG_MAX_FIELDS;
                FieldTable t;
                t.nfields = n;
                for (int f = 0; f < n; ++f) { t.ptr[f] = fields[f0 + f]; t.xloc[f] = xloc[f0 + f]; t.yloc[f] = yloc[f0 + f]; t.sign[f] = sign[f0 + f]; t.item0[f] = 0; }
                t.item0[n] = 0;
                // chunk items (plain or GEN: chunk_plan); TPG_FILL_FUSED=2 (test library) forces the one-thread-per-cell form k_fill_fused
                if (mode != 2) {
                    const ChunkPlan cp = ft == TPG_F64 ? chunk_plan<double>(g, t.ptr, n) : chunk_plan<float>(g, t.ptr, n);
                    const int W = cp.W, r = cp.gen ? Hx % W : 0;
                    FusedVecArgs v{ Nx, Ny, Hx, Hy, Hz, Nz, g.sx, (long long)g.sx * g.sy, cp.gen ? 2 * (Hx / W) + Nx / W : g.sx / W,
                                    cp.gen ? Hx : 2 * Hx / W, 0, 0, r, (Hy + 1) * 2 * r };
                    v.itemsA = (Hy + 1) * v.cpr;
                    v.per_level = v.itemsA + v.itemsS + (Ny + Hy - 1) * v.hc;
                    dim3 gridv((unsigned)(((long long)v.per_level * (Nz + 2 * Hz) + 255) / 256), (unsigned)n);
                    if (ft == TPG_F64) fused_vec_dispatch<double>(gridv, s, t, v, cp);
                    else               fused_vec_dispatch<float>(gridv, s, t, v, cp);
                } else {
                    dim3 grid((unsigned)((items + 255) / 256), (unsigned)n);
                    if (ft == TPG_F64) TPG_LAUNCH(k_fill_fused<double>, grid, dim3(256), s, t, a);
                    else               TPG_LAUNCH(k_fill_fused<float>, grid, dim3(256), s, t, a);
                }
                if ((rc = tpg::launch_status("k_fill_fused"))) return rc;
            }
            return TPG_OK;
        }
    }
    // large fields: zipper (with its corner cells) + periodic x merged into one launch (plain or GEN: chunk_plan); TPG_FILL_MERGED=0 never
    if (north_is_zipper && tpg::config().fill_merged != 0 && Hx > 0 && Hy >= 1 && Hy <= 8 && Nx >= 2 * Hx + 2 && Ny >= 2 * Hy + 2
        && fields && xloc && yloc && sign && nfields >= 1 && (long long)Nz * (Nx + 2 * Hx) < (1ll << 31) - 256) {
        bool valid = true;
        for (int f = 0; f < nfields && valid; ++f)
            valid = fields[f] && (xloc[f] == TPG_CENTER || xloc[f] == TPG_FACE) && (yloc[f] == TPG_CENTER || yloc[f] == TPG_FACE);
        if (valid && !(rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft))) {
            Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
            // one plan for the whole call (all batches)
            const ChunkPlan cp = ft == TPG_F64 ? chunk_plan<double>(g, fields, nfields) : chunk_plan<float>(g, fields, nfields);
            const int W = cp.W, r = cp.gen ? Hx % W : 0;
            MergedArgs a{ Nx, Ny, Hx, Hy, Hz, Nz, g.sx, g.sy, g.plane, cp.gen ? 2 * (Hx / W) + Nx / W : g.sx / W, cp.gen ? Hx : Hx / W, 0,
                          (long long)g.sy * (Nz + 2 * Hz), r, (unsigned)(((long long)Nz * 2 * r + 255) / 256) };
            a.blocksA = (unsigned)(((long long)Nz * a.cprA + 255) / 256);
            hipStream_t s = tpg::as_stream(stream);
            for (int f0 = 0; f0 < nfields; f0 += TPG_MAX_FIELDS) {
                const int n = nfields - f0 < TPG_MAX_FIELDS ? nfields - f0 : TPG_MAX_FIELDS;
                FieldTable t;
                t.nfields = n;
                for (int f = 0; f < n; ++f) { t.ptr[f] = fields[f0 + f]; t.xloc[f] = xloc[f0 + f]; t.yloc[f] = yloc[f0 + f]; t.sign[f] = sign[f0 + f]; t.item0[f] = 0; }
                t.item0[n] = 0;
                rc = (ft == TPG_F64) ? merged_dispatch<double>(t, a, n, Hy, cp, s) : merged_dispatch<float>(t, a, n, Hy, cp, s);
                if (rc) return rc;
            }
            return TPG_OK;
        }
    }
    rc = TPG_OK;
    if (north_is_zipper)
        rc = tpg_zipper_fill(fields, nfields, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, 1, Nz, ft, stream);
    if (rc) return rc;
    return tpg_periodic_x_fill(fields, nfields, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream);
}

int tpg_fill_halo_regions_timed(void* const fields[], int nfields, const int8_t xloc[], const int8_t yloc[],
                                const int32_t sign[], int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                                int north_is_zipper, int ft, void* stream, void* start_event, void* stop_event)
{
    if (nfields > TPG_MAX_FIELDS) { tpg::set_error("timed launch: at most %d fields (one batch)", TPG_MAX_FIELDS); return TPG_ERR_UNSUPPORTED; }
    tpg::ev_start = static_cast<hipEvent_t>(start_event);
    tpg::ev_stop = static_cast<hipEvent_t>(stop_event);
    int rc = tpg_fill_halo_regions(fields, nfields, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, north_is_zipper, ft, stream);
    tpg::ev_start = tpg::ev_stop = nullptr;
    return rc;
}

size_t tpg_y_halo_buffer_elems(int nfields, int Nx, int Nz, int Hx, int Hy, int Hz)
{
    if (nfields < 0 || Nx < 0 || Nz < 0 || Hx < 0 || Hy < 0 || Hz < 0) return 0;
    return (size_t)nfields * (size_t)(Nx + 2 * Hx) * (size_t)Hy * (size_t)(Nz + 2 * Hz);
}

static int pack_common(void* const fields[], int nfields, void* buffer, int side, bool pack,
                       int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    int rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft);
    if (rc) return rc;
    if ((rc = check_fields(fields, nfields))) return rc;
    if (!buffer) { tpg::set_error("null message buffer"); return TPG_ERR_INVALID_ARGUMENT; }
    if (side != 0 && side != 1) { tpg::set_error("side must be 0 (south) or 1 (north)"); return TPG_ERR_INVALID_ARGUMENT; }
    if (nfields > TPG_MAX_FIELDS) { tpg::set_error("at most %d fields per message", TPG_MAX_FIELDS); return TPG_ERR_UNSUPPORTED; }
    if (Hy == 0) return TPG_OK;
    Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
    // 0-based parent row of the slab: pack reads interior rows next to the side, unpack writes halo rows
    int row0 = pack ? (side == 0 ? Hy : Ny) : (side == 0 ? 0 : Ny + Hy);
    const size_t esz = ft == TPG_F64 ? 8 : 4;
    // 16-B chunks of the rows when every row start, the message buffer and every field base sit on the 16-B grid (Float64 with aligned
    // bases: always -- sx is even); otherwise 16-B chunks of the contiguous Hy x sx slabs through element-aligned accesses (k_pack_loose):
    // Float32 rows with sx = 2 mod 4 -- e.g. Nx = 3600 at the reference's model halo 5 -- or bases off the grid
    uintptr_t low = (uintptr_t)buffer | (uintptr_t)((size_t)g.sx * esz);
    PtrTable pt;
    for (int f = 0; f < nfields; ++f) { pt.ptr[f] = fields[f]; low |= (uintptr_t)fields[f]; }
    const bool vec = low % 16 == 0;
    const int W = (int)(16 / esz);
    PackArgs a{ g.sx, g.sy, Nz + 2 * Hz, Hy, row0, g.plane, nfields, vec ? W : 1 };
    long long total = vec ? (long long)nfields * a.nlev * Hy * g.sx / W : (long long)nfields * a.nlev * ((Hy * g.sx + W - 1) / W);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = tpg::as_stream(stream);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (vec) {
        if (pack) hipLaunchKernelGGL((k_pack<u32x4, true>), grid, dim3(256), 0, s, pt, static_cast<u32x4*>(buffer), a);
        else      hipLaunchKernelGGL((k_pack<u32x4, false>), grid, dim3(256), 0, s, pt, static_cast<u32x4*>(buffer), a);
    } else if (ft == TPG_F64) {
        if (pack) hipLaunchKernelGGL((k_pack_loose<double, 2, true>), grid, dim3(256), 0, s, pt, static_cast<double*>(buffer), a);
        else      hipLaunchKernelGGL((k_pack_loose<double, 2, false>), grid, dim3(256), 0, s, pt, static_cast<double*>(buffer), a);
    } else {
        if (pack) hipLaunchKernelGGL((k_pack_loose<float, 4, true>), grid, dim3(256), 0, s, pt, static_cast<float*>(buffer), a);
        else      hipLaunchKernelGGL((k_pack_loose<float, 4, false>), grid, dim3(256), 0, s, pt, static_cast<float*>(buffer), a);
    }
    return tpg::launch_status("k_pack");
}

int tpg_pack_y_halo(void* const fields[], int nfields, void* buffer, int side,
                    int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    return pack_common(fields, nfields, buffer, side, true, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream);
}

int tpg_unpack_y_halo(void* const fields[], int nfields, const void* buffer, int side,
                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    return pack_common(fields, nfields, const_cast<void*>(buffer), side, false, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream);
}

}  // extern "C"

