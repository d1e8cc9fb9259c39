// micro-benchmark: dependent-chain latencies on one wave (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_div(double* out, double a, double b, int n) { double x = a; for (int i = 0; i < n; i++) x = b / (x + 1.0); out[threadIdx.x] = x; }
__global__ void k_sqrt(double* out, double a, int n) { double x = a; for (int i = 0; i < n; i++) x = sqrt(x + 2.0); out[threadIdx.x] = x; }
__global__ void k_fma(double* out, double a, double b, int n) { double x = a; for (int i = 0; i < n; i++) x = x * b + a; out[threadIdx.x] = x; }
__global__ void k_sincos(double* out, double a, int n) { double x = a; for (int i = 0; i < n; i++) x = sin(x) + cos(x); out[threadIdx.x] = x; }
__global__ void k_chase(const int* next, int* out, int n) { int p = threadIdx.x; for (int i = 0; i < n; i++) p = next[p]; out[threadIdx.x] = p; }
__global__ void k_lds(int* out, int n) { __shared__ int s[1024]; for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (i * 37 + 11) & 1023; __syncthreads(); int p = threadIdx.x; for (int i = 0; i < n; i++) p = s[p]; out[threadIdx.x] = p; }
__global__ void k_shfl(double* out, double a, int n) { double x = a + threadIdx.x; for (int i = 0; i < n; i++) x += __shfl_xor(x, 1 << (i % 6), 64); out[threadIdx.x] = x; }
__global__ void k_sync(int* out, int n) { int x = 0; for (int i = 0; i < n; i++) { __syncthreads(); x += i; } out[threadIdx.x] = x; }
int main() {
    double* d; int* di; int* nx;
    hipMalloc(&d, 8192); hipMalloc(&di, 8192);
    const int N = 1 << 20; hipMalloc(&nx, N * 4);
    int* h = new int[N]; for (int i = 0; i < N; i++) h[i] = (int)(((long)i * 7919 + 104729) % N); hipMemcpy(nx, h, N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    auto run = [&](const char* name, auto launch, int threads) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s threads %4d: %8.1f ns/iter  (~%6.0f cycles @2.4GHz)\n", name, threads, ms * 1e6 / n, ms * 1e6 / n * 2.4);
    };
    for (int th : {64, 512}) {
        run("f64 div", [&] { hipLaunchKernelGGL(k_div, 1, th, 0, 0, d, 1.5, 2.5, n); }, th);
        run("f64 sqrt", [&] { hipLaunchKernelGGL(k_sqrt, 1, th, 0, 0, d, 1.5, n); }, th);
        run("f64 fma", [&] { hipLaunchKernelGGL(k_fma, 1, th, 0, 0, d, 1.0000001, 0.9999999, n); }, th);
        run("f64 sincos", [&] { hipLaunchKernelGGL(k_sincos, 1, th, 0, 0, d, 0.3, n); }, th);
        run("gl chase", [&] { hipLaunchKernelGGL(k_chase, 1, th, 0, 0, nx, di, n); }, th);
        run("lds chase", [&] { hipLaunchKernelGGL(k_lds, 1, th, 0, 0, di, n); }, th);
        run("shfl f64", [&] { hipLaunchKernelGGL(k_shfl, 1, th, 0, 0, d, 1.0, n); }, th);
        run("syncthr", [&] { hipLaunchKernelGGL(k_sync, 1, th, 0, 0, di, n); }, th);
    }
    return 0;
}
