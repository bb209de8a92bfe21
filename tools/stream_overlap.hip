// Do kernels on different HIP streams overlap on this box?  (diagnostic for the side-by-side voter lanes)
//   hipcc --offload-arch=gfx950 -O2 tools/stream_overlap.hip -o /tmp/stream_overlap && /tmp/stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (cycles < 0) *sink = 1;
}

static double run(int n_streams, int launches, int grid, int block, long long cycles, std::vector<hipStream_t> &st, int *sink)
{
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < launches; ++i)
        for (int s = 0; s < n_streams; ++s) hipLaunchKernelGGL(spin, dim3(grid), dim3(block), 0, st[s], cycles, sink);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main()
{
    int *sink;
    CK(hipMalloc(&sink, 4));
    std::vector<hipStream_t> st(8);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const long long c = 5000;      // 100 MHz wall clock: 50 us
    for (int grid : {8, 64, 256, 1024})
        for (int ns : {1, 2, 4, 8}) {
            run(ns, 5, grid, 256, c, st, sink);
            const double t = run(ns, 50, grid, 256, c, st, sink);
            printf("grid %4d x 256 threads, 50 us each: %d stream(s) x 50 launches -> %8.1f us  (%.2f x one stream's work)\n", grid, ns, t, t / (50 * 50.0));
        }
    // the same through captured graphs (one graph of 10 launches per stream, replayed 5 times)
    for (int grid : {8, 256})
        for (int ns : {1, 2, 4}) {
            std::vector<hipGraphExec_t> ex(ns);
            for (int s = 0; s < ns; ++s) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st[s], c, sink);
                CK(hipStreamEndCapture(st[s], &g));
                CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
            }
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 5; ++i)
                    for (int s = 0; s < ns; ++s) CK(hipGraphLaunch(ex[s], st[s]));
                CK(hipDeviceSynchronize());
                const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep) printf("graphs: grid %4d, %d stream(s) x 5 replays of 10 launches -> %8.1f us  (%.2f x one stream's work)\n", grid, ns, t, t / (50 * 50.0));
            }
        }
    return 0;
}
