// Does a HIP runtime thread spin while long kernels run on several streams?  (round 6: one unnamed thread used 0.92 of a core during bench.py)
// Eight streams, each: 100 ms kernel, event record, the caller polls the events with sleeps.  Reports the CPU time of the other threads.
// hipcc --offload-arch=gfx950 -O2 -o build/runtime_spin_probe tools/experiments/runtime_spin_probe.hip ;  usage: runtime_spin_probe [mode]
//   mode 0: kernels + events only; 1: + a cross-stream hipStreamWaitEvent chain; 2: + timing events (hipEventElapsedTime); 3: + small D2H copies
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/resource.h>
#include <time.h>
#include <unistd.h>

__global__ void spin_kernel(unsigned long long* out, unsigned iters) {
    unsigned long long a = threadIdx.x;
    for (unsigned i = 0; i < iters; i++) a = a * 6364136223846793005ull + 1442695040888963407ull;
    if (a == 42) out[0] = a;
}
static double cpu_of(int who) {
    rusage r;
    getrusage(who, &r);
    return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int S = 8, ROUNDS = 10;
    unsigned long long *d, *h;
    hipMalloc(&d, 1 << 20);
    hipHostMalloc(&h, 1 << 20, 0);
    hipStream_t st[S], aux;
    hipEvent_t ev[S], e0[S], x[S];
    hipStreamCreate(&aux);
    for (int i = 0; i < S; i++) {
        hipStreamCreate(&st[i]);
        hipEventCreateWithFlags(&ev[i], mode == 2 ? 0 : hipEventDisableTiming);
        hipEventCreate(&e0[i]);
        hipEventCreateWithFlags(&x[i], hipEventDisableTiming);
    }
    spin_kernel<<<64, 64, 0, st[0]>>>(d, 1000);
    hipDeviceSynchronize();
    const double w0 = cpu_of(RUSAGE_SELF), m0 = cpu_of(RUSAGE_THREAD);
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int r = 0; r < ROUNDS; r++) {
        for (int i = 0; i < S; i++) {
            if (mode == 2) hipEventRecord(e0[i], st[i]);
            spin_kernel<<<64, 64, 0, st[i]>>>(d, 3500000);  // ~ 100 ms
            if (mode == 1) {  // hand over to another stream and back, as the commitment scheduler does
                hipEventRecord(x[i], st[i]);
                hipStreamWaitEvent(aux, x[i], 0);
                spin_kernel<<<64, 64, 0, aux>>>(d, 350000);
                hipEventRecord(x[i], aux);
                hipStreamWaitEvent(st[i], x[i], 0);
            }
            if (mode == 3) hipMemcpyAsync(h + 64 * i, d, 512, hipMemcpyDeviceToHost, st[i]);
            hipEventRecord(ev[i], st[i]);
        }
        for (int i = 0; i < S; i++) {
            while (hipEventQuery(ev[i]) != hipSuccess) usleep(200);
            if (mode == 2) { float ms; hipEventElapsedTime(&ms, e0[i], ev[i]); }
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    const double wall = (b.tv_sec - a.tv_sec) + (b.tv_nsec - a.tv_nsec) * 1e-9;
    const double w1 = cpu_of(RUSAGE_SELF), m1 = cpu_of(RUSAGE_THREAD);
    printf("mode %d: wall %.2f s, caller CPU %.3f s, other threads CPU %.3f s (%.0f %% of one core)\n", mode, wall, m1 - m0, (w1 - w0) - (m1 - m0),
           100 * ((w1 - w0) - (m1 - m0)) / wall);
    return 0;
}
