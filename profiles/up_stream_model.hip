// Model of a barrier-free fused up-sample + blend for config #2 (512^2 -> 4096^2 Triangle + 3-step chain on 3 channels):
// a wave owns 256 output columns (a float4 per lane) and a band of rows; per output row it does what the real kernel
// would -- a 3-tap vertical combination of register-held source rows, 12 cross-lane reads (ds_bpermute) for the horizontal
// windows, 12 multiply-adds, 3 blend steps against a streamed operand plane -- and stores the row.  Only the data flow and
// instruction mix are modelled (weights are arbitrary), to see what such a structure reaches before building it.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/up_stream_model.hip -o /tmp/usm && /tmp/usm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PF>
__global__ __launch_bounds__(256) void model(const float *__restrict__ small, uint32_t spitch, const f4 *__restrict__ other, f4 *__restrict__ out,
                                             uint32_t w4, uint32_t h, uint32_t band, float c0, float c1)
{
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t strips = w4 / 64u;
    const uint32_t plane = blockIdx.y;
    const uint32_t sx = wave % strips, by = wave / strips;
    const uint32_t y0 = by * band;
    if (y0 >= h) return;
    const uint32_t y1 = min(y0 + band, h);
    other += (size_t)plane * w4 * h;
    out += (size_t)plane * w4 * h;
    small += (size_t)plane * spitch * (h / 8u + 4u);
    const uint32_t q = sx * 64u + lane;
    // this lane's column of the source strip (34 of the 64 lanes matter) and its three live rows
    const float *scol = small + sx * 32u + min(lane, 35u);
    uint32_t m = y0 / 8u;
    float r0 = scol[(size_t)m * spitch], r1 = scol[(size_t)(m + 1) * spitch], r2 = scol[(size_t)(m + 2) * spitch];
    const int idx0 = (int)((lane >> 1) * 4u), idx1 = idx0 + 4, idx2 = idx0 + 8;  // byte indices for ds_bpermute
    const float wh[3] = { 0.25f + 0.001f * lane, 0.5f, 0.25f - 0.001f * lane };
    f4 ob[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ob[i] = other[(size_t)min(y0 + i, y1 - 1u) * w4 + q];
    for (uint32_t yb = y0; yb < y1; yb += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const uint32_t y = yb + i;
            if (y >= y1) break;
            const f4 o = ob[i];
            if (y + PF < y1) ob[i] = other[(size_t)(y + PF) * w4 + q];
            const uint32_t ph = y & 7u;
            if (ph == 0u && y != y0) {  // the window moves down one source row
                ++m;
                r0 = r1;
                r1 = r2;
                r2 = scol[(size_t)(m + 2) * spitch];
            }
            const float wv0 = 0.125f * (float)(8u - ph), wv1 = 0.5f, wv2 = 0.125f * (float)ph;  // per-phase weights (scalar)
            float t = 0.0f;
            t += r0 * wv0;
            t += r1 * wv1;
            t += r2 * wv2;
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx0, __builtin_bit_cast(int, t)));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx1, __builtin_bit_cast(int, t)));
            const float c = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx2, __builtin_bit_cast(int, t)));
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = 0.0f;
                u += a * (wh[0] + 0.01f * e);
                u += b * (wh[1] - 0.01f * e);
                u += c * (wh[2] + 0.02f * e);
                u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
                v[e] = u;
            }
            f4 x = v + o;          // three blend steps
            x = c0 - x;
            x = x * o;
            (void)c1;
            out[(size_t)y * w4 + q] = x;
        }
    }
}

template <int PF>
static void run(uint32_t band)
{
    const uint32_t w = 4096, h = 4096, planes = 3;
    float *small;
    f4 *other, *out;
    const uint32_t spitch = 576;
    CK(hipMalloc((void **)&small, (size_t)planes * spitch * (h / 8 + 4) * 4));
    CK(hipMalloc((void **)&other, (size_t)planes * w * h * 4));
    CK(hipMalloc((void **)&out, (size_t)planes * w * h * 4));
    CK(hipMemset(small, 0, (size_t)planes * spitch * (h / 8 + 4) * 4));
    CK(hipMemset(other, 0, (size_t)planes * w * h * 4));
    const uint32_t strips = w / 256, bands = (h + band - 1) / band, waves = strips * bands;
    dim3 grid((waves + 3) / 4, planes);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) model<PF><<<grid, 256>>>(small, spitch, other, out, w / 4, h, band, 1.0f, 0.5f);
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) model<PF><<<grid, 256>>>(small, spitch, other, out, w / 4, h, band, 1.0f, 0.5f);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / 20;
    std::printf("PF=%d band=%4u waves/plane=%5u : %6.1f us  (%.0f GB/s of the 403 MB a 1-in-1-out stream moves)\n", PF, band, waves, us, 2.0 * planes * w * h * 4 / us * 1e-3);
    CK(hipFree(small)); CK(hipFree(other)); CK(hipFree(out));
}

int main()
{
    for (uint32_t band : { 32u, 64u, 128u, 256u }) {
        run<4>(band);
        run<8>(band);
    }
    return 0;
}
