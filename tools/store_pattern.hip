// Diagnostic micro-benchmark (not part of the product): how fast can the chip write the feature store in the vectoriser's access
// pattern -- every workgroup writes whole 16 KB rows (one per view, the views 1.6 GB apart) of the sequences it owns -- compared
// with a linear fill?   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0: plain stores, 1: nontemporal, 2: plain + a barrier and ~2 us of LDS busy work between views
__global__ __launch_bounds__(256) void rows_kernel(float *out, int64_t n, int P, int64_t view_stride, int blocked, int64_t seq_stride = 4096)
{
    __shared__ uint32_t junk[4096];
    const int tid = threadIdx.x;
    const int64_t G = gridDim.x;
    const int64_t per = (n + G - 1) / G;
    for (int64_t i = 0; i < per; ++i) {
        const int64_t s = blocked ? (int64_t)blockIdx.x * per + i : (int64_t)blockIdx.x + i * G;
        if (s >= n) break;
        for (int v = 0; v < P; ++v) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 *row = (f4 *)(out + v * view_stride + s * seq_stride);
            const f4 val = {(float)s, (float)v, 1.f, 2.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 1) __builtin_nontemporal_store(val, row + tid + j * 256);
                else row[tid + j * 256] = val;
            }
            if (MODE == 2) {
                for (int r = 0; r < 40; ++r) atomicAdd(&junk[(tid * 17 + r * 97 + (int)s) & 4095], 1u);
                __syncthreads();
            }
        }
    }
}

// one wave of each workgroup stores the whole 16 KB row (16 pieces per lane), the others idle at the barrier: the memory-wave shape
template <int NW>     // storing waves per workgroup (1, 2 or 4)
__global__ __launch_bounds__(256) void rows_by_waves_kernel(float *out, int64_t n, int P, int64_t view_stride)
{
    __shared__ uint32_t junk[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t G = gridDim.x;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int64_t s = blockIdx.x; s < n; s += G) {
        for (int v = 0; v < P; ++v) {
            if (wave < NW) {
                f4 *row = (f4 *)(out + v * view_stride + s * 4096);
                const f4 val = {(float)s, (float)v, 1.f, 2.f};
#pragma unroll
                for (int j = 0; j < 16 / NW; ++j) row[(wave * (16 / NW) + j) * 64 + lane] = val;
            } else {
                for (int r = 0; r < 40; ++r) atomicAdd(&junk[(tid * 17 + r * 97 + (int)s) & 4095], 1u);
            }
            __syncthreads();
        }
    }
}

// persistent workgroups that take the next sequence (or chunk of C sequences) from a global counter instead of a fixed stride
template <int C>
__global__ __launch_bounds__(256) void rows_queue_kernel(float *out, int64_t n, int P, int64_t view_stride, unsigned long long *head)
{
    __shared__ unsigned long long base_s;
    const int tid = threadIdx.x;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (;;) {
        if (tid == 0) base_s = atomicAdd(head, (unsigned long long)C);
        __syncthreads();
        const int64_t base = (int64_t)base_s;
        __syncthreads();
        if (base >= n) break;
        for (int c = 0; c < C && base + c < n; ++c) {
            const int64_t s = base + c;
            for (int v = 0; v < P; ++v) {
                f4 *row = (f4 *)(out + v * view_stride + s * 4096);
                const f4 val = {(float)s, (float)v, 1.f, 2.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) row[tid + j * 256] = val;
            }
        }
    }
}

__global__ void fill_kernel(float4 *out, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main()
{
    const int64_t n = 100000; const int P = 4; const int64_t F = 4096;
    float *out; CK(hipMalloc(&out, (size_t)P * n * F * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char *name, auto launch) {
        launch(); (void)hipDeviceSynchronize(); float best = 1e9f;
        for (int r = 0; r < 6; ++r) { (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-58s %7.3f ms  %6.0f GB/s\n", name, best, (double)P * n * F * 4 / best / 1e6);
    };
    time("linear fill (grid-stride float4)", [&] { hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (float4 *)out, P * n * F / 4); });
    for (int wg : {2, 4, 6, 8}) for (int blocked : {0, 1}) {
        char nm[128];
        snprintf(nm, sizeof nm, "rows, plain stores, %d WG/CU, %s", wg, blocked ? "blocked" : "interleaved");
        time(nm, [&] { hipLaunchKernelGGL(rows_kernel<0>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, n * F, blocked); });
    }
    for (int wg : {4, 8}) for (int blocked : {0, 1}) {
        char nm[128];
        snprintf(nm, sizeof nm, "sequence-major [N][P][F] rows, %d WG/CU, %s", wg, blocked ? "blocked" : "interleaved");
        time(nm, [&] { hipLaunchKernelGGL(rows_kernel<0>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, (int64_t)F, blocked, (int64_t)P * F); });
    }
    time("one-shot workgroups, one 16 KB row each (400k WGs)", [&] { hipLaunchKernelGGL(rows_kernel<0>, dim3((unsigned)(n * P)), dim3(256), 0, 0, out, n * P, 1, (int64_t)0, 0, (int64_t)F); });
    time("one-shot workgroups, one sequence (4 views, view-major) each", [&] { hipLaunchKernelGGL(rows_kernel<0>, dim3((unsigned)n), dim3(256), 0, 0, out, n, P, n * F, 0, (int64_t)F); });
    for (int wg : {3, 4, 6}) {
        char nm[128];
        snprintf(nm, sizeof nm, "rows stored by ONE wave of 4 (others: LDS work), %d WG/CU", wg);
        time(nm, [&] { hipLaunchKernelGGL(rows_by_waves_kernel<1>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, n * F); });
        snprintf(nm, sizeof nm, "rows stored by TWO waves of 4 (others: LDS work), %d WG/CU", wg);
        time(nm, [&] { hipLaunchKernelGGL(rows_by_waves_kernel<2>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, n * F); });
        snprintf(nm, sizeof nm, "rows stored by all FOUR waves, %d WG/CU", wg);
        time(nm, [&] { hipLaunchKernelGGL(rows_by_waves_kernel<4>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, n * F); });
    }
    unsigned long long *head; CK(hipMalloc(&head, 8));
    for (int wg : {4, 8}) {
        char nm[128];
        snprintf(nm, sizeof nm, "rows, persistent WGs + global queue (1 sequence per grab), %d WG/CU", wg);
        time(nm, [&] { (void)hipMemsetAsync(head, 0, 8, 0); hipLaunchKernelGGL(rows_queue_kernel<1>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, n * F, head); });
        snprintf(nm, sizeof nm, "rows, persistent WGs + global queue (4 sequences per grab), %d WG/CU", wg);
        time(nm, [&] { (void)hipMemsetAsync(head, 0, 8, 0); hipLaunchKernelGGL(rows_queue_kernel<4>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, n * F, head); });
    }
    time("rows, nontemporal stores, 4 WG/CU, interleaved", [&] { hipLaunchKernelGGL(rows_kernel<1>, dim3(1024), dim3(256), 0, 0, out, n, P, n * F, 0); });
    for (int wg : {2, 4, 6}) {
        char nm[128];
        snprintf(nm, sizeof nm, "rows + LDS busy work + barrier per view, %d WG/CU", wg);
        time(nm, [&] { hipLaunchKernelGGL(rows_kernel<2>, dim3(256 * wg), dim3(256), 0, 0, out, n, P, n * F, 0); });
    }
    return 0;
}
