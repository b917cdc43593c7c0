// h2d_probe.hip - how fast does hipMemcpyAsync move bytes from a hipHostMalloc'ed buffer to the device, by size and by the
// ALIGNMENT of the source / destination addresses?  (The loader's genome reader copies records out of a pinned read buffer at
// whatever offset their headers leave them: round 5 measured 1.5 GB/s there and left the cause open.)
//   hipcc --offload-arch=gfx950 -O2 h2d_probe.hip -o h2d_probe && ./h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t BUF = (size_t)1 << 30;
    uint8_t *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void **)&h, BUF + 4096, hipHostMallocDefault));
    CK(hipMalloc((void **)&d, BUF + 4096));
    memset(h, 1, BUF + 4096);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t sizes[] = {1250000, (size_t)16 << 20, (size_t)256 << 20};
    const int soffs[] = {0, 16, 47}, doffs[] = {0, 8, 3};
    for (size_t sz : sizes)
        for (int so : soffs)
            for (int dof : doffs) {
                const size_t n = std::max<size_t>(1, std::min<size_t>(400, ((size_t)2 << 30) / sz));
                CK(hipStreamSynchronize(st));
                const double t0 = now();
                for (size_t i = 0; i < n; i++) {
                    const size_t o = (i * (sz + 4096)) % (BUF - sz - 4096);
                    CK(hipMemcpyAsync(d + (o & ~(size_t)63) + dof, h + (o & ~(size_t)63) + so, sz, hipMemcpyHostToDevice, st));
                }
                CK(hipStreamSynchronize(st));
                const double dt = now() - t0;
                printf("{\"bytes\": %zu, \"src_off\": %d, \"dst_off\": %d, \"copies\": %zu, \"GBps\": %.2f, \"us_per_copy\": %.1f}\n", sz, so, dof, n,
                       (double)(sz * n) / dt / 1e9, dt / n * 1e6);
            }
    // reading a file-sized block INTO the pinned buffer by 16 threads (memcpy from ordinary memory: what pread does from the page cache)
    std::vector<uint8_t> src((size_t)256 << 20, 2);
    for (int rep = 0; rep < 2; rep++) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < 16; t++) th.emplace_back([&, t] { memcpy(h + (size_t)t * (16 << 20), src.data() + (size_t)t * (16 << 20), (size_t)16 << 20); });
        for (auto &x : th) x.join();
        printf("{\"memcpy_into_pinned_256MB_16_threads_GBps\": %.2f}\n", 0.268435456 / (now() - t0));
    }
    return 0;
}
