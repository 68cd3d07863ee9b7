// Does v_mfma_f64_16x16x4_f64 accumulate like four sequential FMAs?  (the workgroup LU's
// rank-4 trailing update as one MFMA per 16 x 16 block needs D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c)))) bit
// for bit -- denseGETRF's order -- or the factors stop being the oracle's.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_f64.hip -o tools/ubench_mfma_f64.bin && tools/ubench_mfma_f64.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

// one wavefront: A[16][4], B[4][16], C[16][16] row-major in memory -> D[16][16]
__global__ void __launch_bounds__(64) k_mfma(const double *A, const double *B, const double *C, double *D, int nblk, int layout)
{
    const int l = threadIdx.x;
    for (int t = 0; t < nblk; t++) {
        const double *a = A + t * 64, *b = B + t * 64, *c = C + t * 256;
        double *d = D + t * 256;
        const double av = a[(l % 16) * 4 + l / 16];          // A[i = l % 16][k = l / 16]
        const double bv = b[(l / 16) * 16 + l % 16];         // B[k = l / 16][j = l % 16]
        v4f64 cv;
        // C / D[i][j = l % 16]: i = 4 v + l / 16 (layout 1) or 4 (l / 16) + v (layout 0) -- main() finds out which with integers
        for (int v = 0; v < 4; v++) cv[v] = c[(layout ? 4 * v + l / 16 : 4 * (l / 16) + v) * 16 + l % 16];
        v4f64 dv = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, cv, 0, 0, 0);
        for (int v = 0; v < 4; v++) d[(layout ? 4 * v + l / 16 : 4 * (l / 16) + v) * 16 + l % 16] = dv[v];
    }
}

static uint64_t rng_state = 88172645463325252ull;
static double rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    const double u = (double)(rng_state >> 11) / 9007199254740992.0;            // [0, 1)
    const int e = (int)((rng_state >> 3) % 21) - 10;                              // magnitudes 2^-10 .. 2^10: cancellation happens
    return ldexp(2.0 * u - 1.0, e);
}

int main()
{
    const int nblk = 4096;
    std::vector<double> A(nblk * 64), B(nblk * 64), C(nblk * 256), D(nblk * 256);
    // block 0: small integers (exact in any order) -- checks the operand layout assumed above
    for (int i = 0; i < 64; i++) { A[i] = (i % 7) - 3; B[i] = (i % 5) - 2; }
    for (int i = 0; i < 256; i++) C[i] = (i % 11) - 5;
    for (size_t i = 64; i < A.size(); i++) { A[i] = rnd(); B[i] = rnd(); }
    for (size_t i = 256; i < C.size(); i++) C[i] = rnd();
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    int layout = 0;
    for (; layout < 2; layout++) {
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, nblk, layout);
        hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            double f = C[i * 16 + j];
            for (int k = 0; k < 4; k++) f += A[i * 4 + k] * B[k * 16 + j];
            bad += D[i * 16 + j] != f;
        }
        printf("C / D row of register v in lane l: %s: %ld of 256 integer entries differ\n", layout ? "4 v + l / 16" : "4 (l / 16) + v", bad);
        if (bad == 0) break;
    }
    long bad_layout = 0, n = 0, m_fwd = 0, m_rev = 0, m_sum_first = 0, m_unfused = 0;
    for (int t = 0; t < nblk; t++)
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                const double *a = &A[t * 64 + i * 4], *b = &B[t * 64 + j], c = C[t * 256 + i * 16 + j], d = D[t * 256 + i * 16 + j];
                double f = c, r = c, s = 0.0, u = c;
                for (int k = 0; k < 4; k++) f = fma(a[k], b[k * 16], f);                    // k = 0, 1, 2, 3 onto c
                for (int k = 3; k >= 0; k--) r = fma(a[k], b[k * 16], r);                   // k = 3, 2, 1, 0 onto c
                for (int k = 0; k < 4; k++) s = fma(a[k], b[k * 16], s);                    // products first, c last
                s += c;
                for (int k = 0; k < 4; k++) u = u + a[k] * b[k * 16];                       // unfused
                if (t == 0) { bad_layout += (d != f); continue; }
                n++;
                m_fwd += memcmp(&d, &f, 8) != 0; m_rev += memcmp(&d, &r, 8) != 0;
                m_sum_first += memcmp(&d, &s, 8) != 0; m_unfused += memcmp(&d, &u, 8) != 0;
            }
    printf("v_mfma_f64_16x16x4_f64 on gfx950: operand layout check (integers): %ld of 256 entries differ\n", bad_layout);
    printf("random operands (%ld entries), entries that differ from the host's\n", n);
    printf("   fma chain k = 0,1,2,3 onto c (denseGETRF's order): %ld\n", m_fwd);
    printf("   fma chain k = 3,2,1,0 onto c:                      %ld\n", m_rev);
    printf("   fma chain of the products, c added last:           %ld\n", m_sum_first);
    printf("   unfused multiply-adds k = 0..3 onto c:             %ld\n", m_unfused);
    return 0;
}
