// Probe for the same-node transport of csrc/comm.cpp: can two PROCESSES (here: on the one GPU of a test box) hand a plane over
// with  hipIpcGetMemHandle / hipIpcOpenMemHandle  +  a device-side "produced" counter in POSIX shared memory that the producer's
// stream writes (hipStreamWriteValue64) and the consumer's stream waits for (hipStreamWaitValue64)?
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ipc_probe profiles/ipc_probe.cpp -lrt && /tmp/ipc_probe
//
// The consumer enqueues wait + copy BEFORE the producer has even launched the kernel that fills the plane (the producer sleeps
// first), so correct data on the consumer's side means the stream wait really waited.  Everything is bounded by alarm(90).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_));     \
            _exit(3);                                                                           \
        }                                                                                       \
    } while (0)

static int g_rank = 0;

struct Shared {
    alignas(64) uint64_t produced;           // written by the producer's STREAM
    alignas(64) std::atomic<uint32_t> posted;  // host: handle is in `handle`
    alignas(64) std::atomic<uint32_t> done;    // host: consumer has finished
    hipIpcMemHandle_t handle;
    int can_wait[2];
};

__global__ void fill(float *p, size_t n, float base)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = base + (float)(i & 0xffff);
}

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    alarm(90);
    const size_t bytes = (size_t)64 << 20, n = bytes / 4;
    char name[64];
    std::snprintf(name, sizeof name, "/kc_ipc_probe_%d", (int)getpid());
    int fd = shm_open(name, O_CREAT | O_RDWR | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) {
        perror("shm");
        return 2;
    }
    Shared *sh = (Shared *)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    std::memset((void *)sh, 0, 4096);
    pid_t child = fork();  // before any HIP call
    g_rank = child == 0 ? 1 : 0;
    CK(hipSetDevice(0));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    sh->can_wait[g_rank] = can;
    CK(hipHostRegister(sh, 4096, hipHostRegisterMapped));
    void *dflag = nullptr;
    CK(hipHostGetDevicePointer(&dflag, &sh->produced, 0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int rc = 0;
    if (g_rank == 0) {
        float *buf = nullptr;
        CK(hipMalloc((void **)&buf, bytes));
        CK(hipMemset(buf, 0, bytes));
        CK(hipDeviceSynchronize());
        CK(hipIpcGetMemHandle(&sh->handle, buf));
        sh->posted.store(1, std::memory_order_release);
        usleep(500 * 1000);  // the consumer's wait + copy are queued by now
        for (uint64_t round = 1; round <= 3; ++round) {
            fill<<<1024, 256, 0, s>>>(buf, n, 1000.0f * round);
            CK(hipStreamWriteValue64(s, dflag, round, 0));
            CK(hipStreamSynchronize(s));
            while (sh->done.load(std::memory_order_acquire) < round) usleep(100);
        }
        int st = 0;
        waitpid(child, &st, 0);
        rc = WIFEXITED(st) ? WEXITSTATUS(st) : 9;
        std::printf("can_wait_value: %d %d; consumer exit %d\n", sh->can_wait[0], sh->can_wait[1], rc);
        CK(hipFree(buf));
        shm_unlink(name);
    } else {
        while (!sh->posted.load(std::memory_order_acquire)) usleep(100);
        void *peer = nullptr;
        double t0 = now();
        CK(hipIpcOpenMemHandle(&peer, sh->handle, hipIpcMemLazyEnablePeerAccess));
        double t_open = now() - t0;
        float *mine = nullptr, *host = (float *)std::malloc(bytes);
        CK(hipMalloc((void **)&mine, bytes));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (uint64_t round = 1; round <= 3; ++round) {
            CK(hipStreamWaitValue64(s, dflag, round, hipStreamWaitValueGte, ~0ull));
            CK(hipEventRecord(e0, s));
            CK(hipMemcpyAsync(mine, peer, bytes, hipMemcpyDeviceToDevice, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(host, mine, bytes, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < n; ++i) bad += host[i] != 1000.0f * round + (float)(i & 0xffff);
            std::printf("round %llu: open %.3f ms, copy of 64 MiB from the peer mapping %.3f ms (%.1f GB/s), mismatches %zu\n",
                        (unsigned long long)round, t_open * 1e3, ms, bytes / ms / 1e6, bad);
            if (bad) rc = 4;
            sh->done.store((uint32_t)round, std::memory_order_release);
        }
        CK(hipIpcCloseMemHandle(peer));
        CK(hipFree(mine));
        std::fflush(stdout);
        _exit(rc);
    }
    return rc;
}
