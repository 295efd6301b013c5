/* c_host.c -- a plain C99 host of libtripolar_hip.so: no Python, no C++, no torch.
 *
 * What a Julia `ccall`, a cgo or a JNI stub does, spelled out in C: allocate device memory with the HIP C API, call the
 * C ABI of include/tripolar_hip.h, copy a few values back.  Reproduces the reference's README transcript
 * (/root/reference/README.md:52-60: TripolarGrid(size = (60, 30, 1))) and the zipper test of
 * test/test_zipper_boundary_conditions.jl:25-45 (fields of ones on a 10 x 10 x 1 grid).
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host.c \
 *       -Lorthogonalsphericalshellgrids.jl_amd -ltripolar_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/orthogonalsphericalshellgrids.jl_amd -Wl,-rpath,/opt/rocm/lib -o c_host && ./c_host
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "tripolar_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define TPGCHECK(x) do { int s_ = (x); if (s_ != TPG_OK) { fprintf(stderr, "tpg status %d (%s): %s\n", s_, tpg_status_string(s_), tpg_last_error()); return 3; } } while (0)

int main(void)
{
    /* ---- TripolarGrid(size = (60, 30, 1)): defaults of src/tripolar_grid.jl:59-66 ---- */
    tpg_params p = { 60, 30, 1, 4, 4, 4, -80.0, 55.0, 70.0, 6371.0e3, TPG_F64, 1, 30, 0 };
    const size_t sx = (size_t)p.Nx + 2 * p.Hx, sy = (size_t)p.Ny + 2 * p.Hy, n2 = sx * sy;
    void *arrays[TPG_NUM_ARRAYS], *workspace;
    int q;
    for (q = 0; q < TPG_NUM_ARRAYS; ++q) HIPCHECK(hipMalloc(&arrays[q], n2 * sizeof(double)));
    const size_t wbytes = tpg_build_grid_workspace_bytes(&p);
    HIPCHECK(hipMalloc(&workspace, wbytes));
    TPGCHECK(tpg_build_grid(&p, arrays, workspace, wbytes, NULL));
    HIPCHECK(hipDeviceSynchronize());

    double *lam_ff = (double *)malloc(n2 * sizeof(double)), *phi_ff = (double *)malloc(n2 * sizeof(double));
    double *dx_ff = (double *)malloc(n2 * sizeof(double)), *dy_ff = (double *)malloc(n2 * sizeof(double));
    HIPCHECK(hipMemcpy(lam_ff, arrays[TPG_LAMBDA_FF], n2 * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(phi_ff, arrays[TPG_PHI_FF], n2 * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(dx_ff, arrays[TPG_DX_FF], n2 * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(dy_ff, arrays[TPG_DY_FF], n2 * sizeof(double), hipMemcpyDeviceToHost));
    /* A[i, j] at (i + Hx - 1) + sx * (j + Hy - 1); the README's "centered at (lambda, phi)" is node (Nx/2+1, Ny/2+1) */
    {
        const int i = p.Nx / 2 + 1, j = p.Ny / 2 + 1;
        const size_t c = (size_t)(i + p.Hx - 1) + sx * (size_t)(j + p.Hy - 1);
        double dxmin = 1e300, dxmax = 0, dymin = 1e300, dymax = 0;
        int ii, jj;
        for (jj = 1; jj <= p.Ny; ++jj)
            for (ii = 1; ii <= p.Nx; ++ii) {
                const size_t k = (size_t)(ii + p.Hx - 1) + sx * (size_t)(jj + p.Hy - 1);
                if (dx_ff[k] < dxmin) dxmin = dx_ff[k];
                if (dx_ff[k] > dxmax) dxmax = dx_ff[k];
                if (dy_ff[k] < dymin) dymin = dy_ff[k];
                if (dy_ff[k] > dymax) dymax = dy_ff[k];
            }
        printf("grid 60x30x1: center node (lambda, phi) = (%.6g, %.6g); dx_ff in [%.6g, %.6g] m; dy_ff in [%.6g, %.6g] m\n",
               lam_ff[c], phi_ff[c], dxmin, dxmax, dymin, dymax);
    }

    /* ---- fill_halo_regions!(u) with u = 1 on a (Face, Center, Center) field of a 10 x 10 x 1 grid: zipper sign -1 ---- */
    {
        const int Nx = 10, Ny = 10, Nz = 1, H = 4;
        const size_t fsx = Nx + 2 * H, fsy = Ny + 2 * H, n3 = fsx * fsy * (Nz + 2 * H);
        double *h = (double *)calloc(n3, sizeof(double));
        void *u;
        void *fields[1];
        const int8_t xloc[1] = { TPG_FACE }, yloc[1] = { TPG_CENTER };
        const int32_t sign[1] = { -1 };
        int i, j;
        for (j = 1; j <= Ny; ++j)
            for (i = 1; i <= Nx; ++i) h[(size_t)(i + H - 1) + fsx * ((size_t)(j + H - 1) + fsy * (size_t)H)] = 1.0;   /* set!(u, 1), level k = 1 */
        HIPCHECK(hipMalloc(&u, n3 * sizeof(double)));
        HIPCHECK(hipMemcpy(u, h, n3 * sizeof(double), hipMemcpyHostToDevice));
        fields[0] = u;
        TPGCHECK(tpg_fill_halo_regions(fields, 1, xloc, yloc, sign, Nx, Ny, Nz, H, H, H, 1, TPG_F64, NULL));
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(h, u, n3 * sizeof(double), hipMemcpyDeviceToHost));
#define U(i, j) h[(size_t)((i) + H - 1) + fsx * ((size_t)((j) + H - 1) + fsy * (size_t)H)]
        printf("zipper 10x10x1, u = 1, sign -1: u[2, Ny+1] = %g, u[1, Ny+1] = %g, u[Nx+1, Ny+1] = %g, u[Nx/2+1, Ny+4] = %g\n",
               U(2, Ny + 1), U(1, Ny + 1), U(Nx + 1, Ny + 1), U(Nx / 2 + 1, Ny + 4));
        /* the same fill through the distributed entry point on a chain of ONE latitude band (no seam, no communicator):
           zipper -> periodic x -> (no exchange).  The folded pole point of row Ny flips sign on every fill (a quirk of the
           reference, src/zipper_boundary_condition.jl:90,102), everything else is idempotent */
        TPGCHECK(tpg_fill_halo_regions_distributed(NULL, 0, 1, fields, 1, xloc, yloc, sign, NULL, NULL, NULL, NULL,
                                                   Nx, Ny, Nz, H, H, H, TPG_F64, NULL));
        HIPCHECK(hipDeviceSynchronize());
        {
            double *h2 = (double *)malloc(n3 * sizeof(double));
            size_t q2, diff = 0;
            HIPCHECK(hipMemcpy(h2, u, n3 * sizeof(double), hipMemcpyDeviceToHost));
            for (q2 = 0; q2 < n3; ++q2) diff += (h2[q2] != h[q2]);
            printf("distributed entry point, one band: %lu cells differ from the serial fill (the pole point u[Nx/2+1, Ny]); rccl %s\n",
                   (unsigned long)diff, tpg_comm_available() == TPG_OK ? "bound" : "absent");
            free(h2);
        }
        /* ... and through its pipelined form (tpg_fill_halo_regions_distributed_pipelined: the seam exchange in stages of k fields on a
           second stream).  A one-band chain has no seam, so no communicator, buffers or second stream are needed: it is the serial fill
           again, and the pole point flips back */
        TPGCHECK(tpg_fill_halo_regions_distributed_pipelined(NULL, 0, 1, fields, 1, xloc, yloc, sign, NULL, NULL, NULL, NULL,
                                                             Nx, Ny, Nz, H, H, H, TPG_F64, NULL, NULL, 1));
        HIPCHECK(hipDeviceSynchronize());
        {
            double *h3 = (double *)malloc(n3 * sizeof(double));
            size_t q3, diff3 = 0;
            HIPCHECK(hipMemcpy(h3, u, n3 * sizeof(double), hipMemcpyDeviceToHost));
            for (q3 = 0; q3 < n3; ++q3) diff3 += (h3[q3] != h[q3]);
            printf("pipelined entry point, one band: %lu cells differ from the serial fill\n", (unsigned long)diff3);
            free(h3);
        }
        /* an argument error comes back as a status + message, never as an exception across the boundary */
        printf("odd Nlambda -> status %d: %s\n",
               tpg_fill_halo_regions(fields, 1, xloc, yloc, sign, 11, Ny, Nz, H, H, H, 1, TPG_F64, NULL), tpg_last_error());
        (void)hipFree(u);
        free(h);
    }
    for (q = 0; q < TPG_NUM_ARRAYS; ++q) (void)hipFree(arrays[q]);
    (void)hipFree(workspace);
    free(lam_ff); free(phi_ff); free(dx_ff); free(dy_ff);
    printf("libtripolar_hip version %d\n", tpg_version());
    return 0;
}
