// micro-benchmark: throughput of LDS atomics on gfx950 (float add vs integer add vs plain write), random addresses.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomic.hip -o gpurun_out/lds_atomic && ./gpurun_out/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ addr, int iters, float* out) {
    __shared__ float acc[16384];
    unsigned* acci = reinterpret_cast<unsigned*>(acc);
    for (int i = threadIdx.x; i < 16384; i += 256) acc[i] = 0.f;
    __syncthreads();
    int a = addr[blockIdx.x * 256 + threadIdx.x];
    float v = 1.0f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) atomicAdd(acc + a, v);
        else if (MODE == 1) atomicAdd(acci + a, (unsigned)it + 1u);
        else if (MODE == 2) acc[a] = v;
        else { const float o = acc[a]; acc[a] = o + v; }
        a = (a * 1103515245 + 12345) & 16383;
        v += 1.0f;
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 16384; i += 256) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 1024, iters = 2000;
    std::vector<int> h(blocks * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int)((i * 2654435761u) >> 7) & 16383;
    int* d; float* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "read+write"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, o);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, o);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, o);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-14s %8.3f ms  %7.1f G lane-ops/s  (%.2f lane-ops/clk/CU at 2.4 GHz)\n", names[m], ms,
                            blocks * 256.0 * iters / ms * 1e-6, blocks * 256.0 * iters / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
