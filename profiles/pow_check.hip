// Accuracy of the chain kernel's Mix(Pow) fast path (positive finite base, finite exponent) against the
// f64 pow routine rounded to f32 (the correctly rounded power in all but ~2^-28 of cases).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/pow_check.hip -o /tmp/powc && /tmp/powc
// Reports, per sampled region, how many of 2^32 random (a, b) pairs differ and by how many ulp at most.
// The function under test is a verbatim copy of pow_positive() in kanter_core_amd/csrc/kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

#include "../kanter_core_amd/csrc/pow_positive.inc"

static __device__ __forceinline__ uint64_t splitmix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static __device__ __forceinline__ int ulp_diff(float x, float y)
{
    if (__float_as_uint(x) == __float_as_uint(y)) return 0;
    if (x != x || y != y) return (x != x && y != y) ? 0 : 1 << 30;
    int ix = __float_as_int(x), iy = __float_as_int(y);
    ix = ix < 0 ? (int)0x80000000 - ix : ix;
    iy = iy < 0 ? (int)0x80000000 - iy : iy;
    const long long d = (long long)ix - iy;
    return (int)(d < 0 ? -d : d);
}

// mode 0: a, b uniform in (0, 1] / [0, 4)          (image data)
// mode 1: a any positive finite f32 bit pattern, b = +-2^[-20, 8) x mantissa
// mode 2: a = 1 +- tiny, b large (|b log2 a| up to ~150)
__global__ void check(uint64_t n, int mode, uint64_t seed, unsigned long long *diff, int *maxd, uint32_t *worst)
{
    __shared__ double lds[KC_POW_TABLE_DOUBLES];
    const PowCtx tab = pow_setup(lds);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = splitmix(seed + i), r2 = splitmix(r);
        float a, b;
        if (mode == 0) {
            a = ((uint32_t)(r >> 40) + 1) * 0x1p-24f;
            b = (uint32_t)(r2 >> 40) * 0x1p-22f;
        } else if (mode == 1) {
            uint32_t ua = (uint32_t)(r >> 33) % 0x7F800000u;
            if (ua == 0) ua = 1;
            a = __uint_as_float(ua);
            const uint32_t eb = 127 - 20 + (uint32_t)((r2 >> 40) % 28);
            b = __uint_as_float(((uint32_t)(r2 & 1) << 31) | (eb << 23) | ((uint32_t)(r2 >> 1) & 0x7FFFFFu));
        } else {
            a = __uint_as_float(0x3F800000u + (int)((r >> 40) % 4096) - 2048);
            b = (float)((double)(int64_t)(r2 >> 20) * 0x1p-44 * 2.0e8 - 1.0e8);
        }
        const float want = (float)pow((double)a, (double)b);
        const float got = pow_positive(a, b, tab);
        const int d = ulp_diff(got, want);
        if (d) {
            atomicAdd(diff, 1ull);
            if (atomicMax(maxd, d) < d) { worst[0] = __float_as_uint(a); worst[1] = __float_as_uint(b); }
        }
    }
}

int main()
{
    unsigned long long *diff;
    int *maxd;
    uint32_t *worst;
    CK(hipMalloc((void **)&diff, 8));
    CK(hipMalloc((void **)&maxd, 4));
    CK(hipMalloc((void **)&worst, 8));
    const char *names[] = { "a in (0,1], b in [0,4)", "any positive finite a, |b| in [2^-20, 2^8)", "a = 1 +- 2048 ulp, |b| < 1e8" };
    int rc = 0;
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(diff, 0, 8));
        CK(hipMemset(maxd, 0, 4));
        CK(hipMemset(worst, 0, 8));
        const uint64_t n = 1ull << 32;
        check<<<8192, 256>>>(n, mode, 0xC0FFEE00ull + mode, diff, maxd, worst);
        CK(hipDeviceSynchronize());
        unsigned long long hd;
        int hm;
        uint32_t hw[2];
        CK(hipMemcpy(&hd, diff, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hm, maxd, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hw, worst, 8, hipMemcpyDeviceToHost));
        std::printf("%-46s %llu pairs: %llu differ from round(pow_f64), max %d ulp (a=0x%08x b=0x%08x)\n", names[mode],
                    (unsigned long long)n, hd, hm, hw[0], hw[1]);
        rc |= hm > 1;
    }
    return rc;
}
