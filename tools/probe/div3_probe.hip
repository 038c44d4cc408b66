// Is  q = fma(fma(-(a*y), b, a), y, a*y)  with  y = 1.0f / b  (one IEEE division shared by many numerators) bit-identical to the IEEE division a / b?
// (Markstein's correction step: exact in round-to-nearest when y is the correctly rounded reciprocal and no scaling is needed.)
// Counts mismatches over 2^32-ish pseudo-random (a, b) pairs in the ranges the attention epilogue sees (O accumulators / softmax sums) and wider.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ unsigned rnd(unsigned long long& s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(s >> 32); }
__global__ void probe(unsigned long long seed, int mode, unsigned long long* bad, float* ex, unsigned long long* bad0) {
    unsigned long long s = seed + (blockIdx.x * 256ULL + threadIdx.x) * 0x9E3779B97F4A7C15ULL;
    unsigned long long nb = 0, nb0 = 0;
    for (int it = 0; it < 4096; ++it) {
        float b, a[8];
        if (mode == 0) {            // b in [2^-2, 2^18) with random mantissa, a in +-[2^-20, 2^20)
            b = __uint_as_float(((125u + rnd(s) % 20u) << 23) | (rnd(s) & 0x7FFFFFu));
            for (int k = 0; k < 8; ++k) a[k] = __uint_as_float(((rnd(s) & 1u) << 31) | ((107u + rnd(s) % 40u) << 23) | (rnd(s) & 0x7FFFFFu));
        } else {                    // arbitrary normal floats of moderate exponent (|e| <= 60): no overflow / underflow of the quotient
            b = __uint_as_float(((67u + rnd(s) % 120u) << 23) | (rnd(s) & 0x7FFFFFu));
            for (int k = 0; k < 8; ++k) a[k] = __uint_as_float(((rnd(s) & 1u) << 31) | ((67u + rnd(s) % 120u) << 23) | (rnd(s) & 0x7FFFFFu));
        }
        const float y = 1.0f / b;
        for (int k = 0; k < 8; ++k) {
            const float ref = a[k] / b;
            const float q0 = a[k] * y;
            if (__float_as_uint(q0) != __float_as_uint(ref)) ++nb0;          // positive control: the plain product a * (1 / b) is NOT the quotient
            const float r = __builtin_fmaf(-q0, b, a[k]);
            const float q = __builtin_fmaf(r, y, q0);
            if (__float_as_uint(q) != __float_as_uint(ref)) { if (nb == 0 && ex) { ex[0] = a[k]; ex[1] = b; ex[2] = ref; ex[3] = q; } ++nb; }
        }
    }
    if (nb) atomicAdd(bad, nb);
    if (nb0) atomicAdd(bad0, nb0);
}
int main() {
    unsigned long long *bad, *bad0; float* ex;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&bad0, 8); (void)hipMalloc(&ex, 16);
    for (int mode = 0; mode < 2; ++mode) {
        (void)hipMemset(bad, 0, 8); (void)hipMemset(bad0, 0, 8); (void)hipMemset(ex, 0, 16);
        for (int rep = 0; rep < 8; ++rep) hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, 12345ULL + rep * 977ULL + mode * 31ULL, mode, bad, ex, bad0);
        (void)hipDeviceSynchronize();
        unsigned long long h, h0; float he[4];
        (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&h0, bad0, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches of %llu quotients", mode, h, 8ULL * 4096 * 256 * 4096 * 8);
        if (h) printf("   e.g. a = %.9g b = %.9g: a / b = %.9g, corrected product = %.9g", he[0], he[1], he[2], he[3]);
        printf("   (positive control: a * (1 / b) alone differs from a / b in %llu of them)\n", h0);
    }
    return 0;
}
