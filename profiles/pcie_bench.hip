// Host-boundary copy strategies for a 64 MiB RGBA8 / 256 MiB f32 result (what kc_image_to_u8 / kc_image_to_f32 hand
// back): plain hipMemcpy into pageable memory, hipHostRegister around the caller's buffer, pinned staging + memcpy.
//   hipcc --offload-arch=gfx950 -O2 profiles/pcie_bench.hip -o /tmp/pcie && /tmp/pcie
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    for (size_t bytes : { (size_t)64 << 20, (size_t)256 << 20 }) {
        char *dev;
        CK(hipMalloc((void **)&dev, bytes));
        CK(hipMemset(dev, 1, bytes));
        char *host = (char *)std::malloc(bytes);
        std::memset(host, 0, bytes);  // touch the pages
        hipStream_t s;
        CK(hipStreamCreate(&s));
        for (int dir = 0; dir < 2; ++dir) {  // 0: D2H, 1: H2D
            auto copy = [&](void *h, size_t n, size_t off) {
                if (dir == 0) CK(hipMemcpyAsync(h, dev + off, n, hipMemcpyDeviceToHost, s));
                else CK(hipMemcpyAsync(dev + off, h, n, hipMemcpyHostToDevice, s));
            };
            double best[3] = { 1e9, 1e9, 1e9 };
            for (int rep = 0; rep < 4; ++rep) {
                double t0 = now();
                copy(host, bytes, 0);
                CK(hipStreamSynchronize(s));
                best[0] = std::min(best[0], now() - t0);
                t0 = now();
                CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
                copy(host, bytes, 0);
                CK(hipStreamSynchronize(s));
                CK(hipHostUnregister(host));
                best[1] = std::min(best[1], now() - t0);
                // pinned double buffer, 8 MiB chunks, memcpy on this thread overlapped with the next DMA
                static char *pin[2] = { nullptr, nullptr };
                const size_t chunk = (size_t)8 << 20;
                if (!pin[0]) { CK(hipHostMalloc((void **)&pin[0], chunk)); CK(hipHostMalloc((void **)&pin[1], chunk)); }
                hipEvent_t ev[2];
                CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
                t0 = now();
                const size_t nch = (bytes + chunk - 1) / chunk;
                if (dir == 0) {
                    copy(pin[0], std::min(chunk, bytes), 0);
                    CK(hipEventRecord(ev[0], s));
                    for (size_t c = 0; c < nch; ++c) {
                        if (c + 1 < nch) {
                            copy(pin[(c + 1) & 1], std::min(chunk, bytes - (c + 1) * chunk), (c + 1) * chunk);
                            CK(hipEventRecord(ev[(c + 1) & 1], s));
                        }
                        CK(hipEventSynchronize(ev[c & 1]));
                        std::memcpy(host + c * chunk, pin[c & 1], std::min(chunk, bytes - c * chunk));
                    }
                } else {
                    for (size_t c = 0; c < nch; ++c) {
                        if (c >= 2) CK(hipEventSynchronize(ev[c & 1]));
                        std::memcpy(pin[c & 1], host + c * chunk, std::min(chunk, bytes - c * chunk));
                        copy(pin[c & 1], std::min(chunk, bytes - c * chunk), c * chunk);
                        CK(hipEventRecord(ev[c & 1], s));
                    }
                    CK(hipStreamSynchronize(s));
                }
                best[2] = std::min(best[2], now() - t0);
                CK(hipEventDestroy(ev[0])); CK(hipEventDestroy(ev[1]));
            }
            std::printf("%3zu MiB %s: pageable %.2f ms (%.1f GB/s) | register+copy+unregister %.2f ms (%.1f GB/s) | pinned staging %.2f ms (%.1f GB/s)\n",
                        bytes >> 20, dir ? "H2D" : "D2H", best[0] * 1e3, bytes / best[0] / 1e9, best[1] * 1e3, bytes / best[1] / 1e9,
                        best[2] * 1e3, bytes / best[2] / 1e9);
        }
        CK(hipFree(dev));
        std::free(host);
    }
    return 0;
}
