// Where does the per-process start-up time of a HIP program go?  (drop-in CLI latency floor)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int *p) { *p = 1; }
int main() {
    double t0 = now();
    int c = 0; hipGetDeviceCount(&c);
    double t1 = now();
    hipSetDevice(0);
    double t2 = now();
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    double t3 = now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t4 = now();
    int *d; hipMalloc(&d, 4);
    double t5 = now();
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s, d); hipStreamSynchronize(s);
    double t6 = now();
    printf("count %.3f setdevice %.3f props %.3f stream %.3f malloc %.3f first kernel %.3f (%s, %d CUs)\n", t1 - t0, t2 - t1, t3 - t2,
           t4 - t3, t5 - t4, t6 - t5, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}
