// Exhaustive / sampled checks of the hand-sequenced IEEE operations used by height_to_normal_kernel's
// fast path against the compiler's correctly rounded ones (-fhip-fp32-correctly-rounded-divide-sqrt).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         profiles/exact_math_check.hip -o /tmp/emc && /tmp/emc
//  * sqrt: every normal f32 in [2^-96, 2^100] (the kernel uses the sequence for [2^-94, 2^18]);
//  * shared-denominator division: 2^34 (a, b) pairs drawn from the bounds the kernel establishes.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

static __device__ __forceinline__ float sqrt_normal(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float h0 = y * 0.5f;
    const float e = __builtin_fmaf(-h0, s0, 0.5f);
    const float h = __builtin_fmaf(h0, e, h0);
    const float s = __builtin_fmaf(s0, e, s0);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}

struct SharedDenominator { float nb, r; };
static __device__ __forceinline__ SharedDenominator shared_denominator(float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    return { -b, __builtin_fmaf(e, r0, r0) };
}
static __device__ __forceinline__ float divide_by(const SharedDenominator &d, float a)
{
    const float m = a * d.r;
    const float f2 = __builtin_fmaf(d.nb, m, a);
    const float f3 = __builtin_fmaf(f2, d.r, m);
    const float f4 = __builtin_fmaf(d.nb, f3, a);
    const float q = __builtin_fmaf(f4, d.r, f3);
    return __builtin_copysignf(q, a);
}

__global__ void check_sqrt(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first_bad)
{
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = sqrt_normal(x), b = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(b)) {
            if (atomicAdd(bad, 1ull) == 0) *first_bad = (uint32_t)i;
        }
    }
}

static __device__ __forceinline__ uint64_t splitmix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// a: sign, exponent in [ea_lo, ea_hi], random mantissa; b: positive, exponent in [eb_lo, eb_hi]
__global__ void check_div(uint64_t n, int ea_lo, int ea_hi, int eb_lo, int eb_hi, uint64_t seed, unsigned long long *bad,
                          uint32_t *first_bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = splitmix(seed + i), r2 = splitmix(r);
        const uint32_t ea = (uint32_t)(ea_lo + (int)((r >> 40) % (uint64_t)(ea_hi - ea_lo + 1)) + 127);
        const uint32_t eb = (uint32_t)(eb_lo + (int)((r2 >> 40) % (uint64_t)(eb_hi - eb_lo + 1)) + 127);
        const float a = __uint_as_float(((uint32_t)(r & 1) << 31) | (ea << 23) | ((uint32_t)(r >> 1) & 0x7FFFFFu));
        const float b = __uint_as_float((eb << 23) | ((uint32_t)(r2 >> 1) & 0x7FFFFFu));
        const float q1 = divide_by(shared_denominator(b), a), q2 = a / b;
        if (__float_as_uint(q1) != __float_as_uint(q2)) {
            if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = __float_as_uint(a); first_bad[1] = __float_as_uint(b); }
        }
    }
}

// The vectorised kernel's variant: the reciprocal of n = sqrt_normal(x) seeded by rsq(x) instead of rcp(n).
static __device__ __forceinline__ SharedDenominator sqrt_denominator(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float h0 = y * 0.5f;
    const float e = __builtin_fmaf(-h0, s0, 0.5f);
    const float h = __builtin_fmaf(h0, e, h0);
    const float s = __builtin_fmaf(s0, e, s0);
    const float d = __builtin_fmaf(-s, s, x);
    const float n = __builtin_fmaf(d, h, s);
    const float er = __builtin_fmaf(-n, y, 1.0f);
    return { -n, __builtin_fmaf(er, y, y) };
}

// a: sign, exponent in [ea_lo, ea_hi]; x: positive, exponent in [ex_lo, ex_hi]; checks a / sqrtf(x)
__global__ void check_div_sqrt(uint64_t n, int ea_lo, int ea_hi, int ex_lo, int ex_hi, uint64_t seed, unsigned long long *bad,
                               uint32_t *first_bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = splitmix(seed + i), r2 = splitmix(r);
        const uint32_t ea = (uint32_t)(ea_lo + (int)((r >> 40) % (uint64_t)(ea_hi - ea_lo + 1)) + 127);
        const uint32_t ex = (uint32_t)(ex_lo + (int)((r2 >> 40) % (uint64_t)(ex_hi - ex_lo + 1)) + 127);
        const float a = __uint_as_float(((uint32_t)(r & 1) << 31) | (ea << 23) | ((uint32_t)(r >> 1) & 0x7FFFFFu));
        const float x = __uint_as_float((ex << 23) | ((uint32_t)(r2 >> 1) & 0x7FFFFFu));
        const float q1 = divide_by(sqrt_denominator(x), a), q2 = a / sqrtf(x);
        if (__float_as_uint(q1) != __float_as_uint(q2)) {
            if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = __float_as_uint(a); first_bad[1] = __float_as_uint(x); }
        }
    }
}

int main()
{
    unsigned long long *bad;
    uint32_t *first;
    CK(hipMalloc((void **)&bad, 8));
    CK(hipMalloc((void **)&first, 8));
    unsigned long long hbad = 0;
    uint32_t hfirst[2] = { 0, 0 };
    auto reset = [&] { CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 8)); };
    auto fetch = [&] { CK(hipDeviceSynchronize()); CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hfirst, first, 8, hipMemcpyDeviceToHost)); };

    const uint32_t lo = (uint32_t)(127 - 96) << 23, hi = (uint32_t)(127 + 100) << 23;
    reset();
    check_sqrt<<<4096, 256>>>(lo, hi, bad, first);
    fetch();
    std::printf("sqrt_normal vs sqrtf over [2^-96, 2^100): %llu values, %llu mismatches (first 0x%08x)\n",
                (unsigned long long)(hi - lo), hbad, hfirst[0]);
    int rc = hbad != 0;

    struct Range { const char *what; int ea_lo, ea_hi, eb_lo, eb_hi; } ranges[] = {
        { "pdx, tz0 / n1  (a in [2^-40, 2^7], b in [2^-16, 2^8])", -40, 7, -16, 8 },
        { "cross / |cross| (a in [2^-90, 2^0], b in [2^-48, 2^1])", -90, 0, -48, 1 },
        { "quotients near 1 (a, b in [2^-3, 2^3])", -3, 3, -3, 3 },
    };
    for (auto &rg : ranges) {
        reset();
        const uint64_t n = 1ull << 34;
        check_div<<<8192, 256>>>(n, rg.ea_lo, rg.ea_hi, rg.eb_lo, rg.eb_hi, 0x5EED0000ull + (uint64_t)rg.ea_lo * 131, bad, first);
        fetch();
        std::printf("divide_by vs a / b, %s: %llu pairs, %llu mismatches (first a=0x%08x b=0x%08x)\n", rg.what,
                    (unsigned long long)n, hbad, hfirst[0], hfirst[1]);
        rc |= hbad != 0;
    }
    struct RangeS { const char *what; int ea_lo, ea_hi, ex_lo, ex_hi; } sranges[] = {
        { "pdx, tz0 / sqrt(q)  (a in [2^-40, 2^7], q in [2^-32, 2^15])", -40, 7, -32, 15 },
        { "cross / sqrt(|cross|^2) (a in [2^-90, 2^0], x in [2^-94, 2^2])", -90, 0, -94, 2 },
        { "quotients near 1 (a in [2^-3, 2^3], x in [2^-6, 2^6])", -3, 3, -6, 6 },
    };
    for (auto &rg : sranges) {
        reset();
        const uint64_t n = 1ull << 34;
        check_div_sqrt<<<8192, 256>>>(n, rg.ea_lo, rg.ea_hi, rg.ex_lo, rg.ex_hi, 0x5EED1000ull + (uint64_t)rg.ea_lo * 131, bad, first);
        fetch();
        std::printf("divide_by(sqrt_denominator(x)) vs a / sqrtf(x), %s: %llu pairs, %llu mismatches (first a=0x%08x x=0x%08x)\n", rg.what,
                    (unsigned long long)n, hbad, hfirst[0], hfirst[1]);
        rc |= hbad != 0;
    }
    return rc;
}
