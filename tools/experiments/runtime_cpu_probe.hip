// How much host CPU the HIP runtime's own threads spend per operation (round 6: 0.116 CPU-seconds per proof went to unnamed threads).
// Each leg issues N operations on one stream, waits for each with an event the way prover.hip's stream_wait does (query, then sleep), and
// reports the process's CPU time minus the calling thread's.   hipcc --offload-arch=gfx950 -O2 -o build/runtime_cpu_probe tools/experiments/runtime_cpu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
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
static void wait_sleeping(hipEvent_t e) {
    for (int i = 0; i < 4; i++)
        if (hipEventQuery(e) == hipSuccess) return;
    unsigned us = 20;
    while (hipEventQuery(e) != hipSuccess) {
        usleep(us);
        if (us < 200) us *= 2;
    }
}

int main() {
    const int N = 2000;
    unsigned long long *d, *h;
    hipMalloc(&d, 1 << 20);
    hipHostMalloc(&h, 1 << 20, 0);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t ev, evb;
    hipEventCreate(&ev);
    hipEventCreateWithFlags(&evb, hipEventBlockingSync);
    spin_kernel<<<256, 64, 0, st>>>(d, 1000);
    hipStreamSynchronize(st);
    struct Leg { const char* name; int kind; } legs[] = {
        {"kernel (1 ms) + event, query/sleep wait", 0}, {"kernel (20 us) + event, query/sleep wait", 1}, {"D2H copy 512 B + event, query/sleep wait", 2},
        {"H2D copy 512 B + event, query/sleep wait", 3}, {"D2H copy 4 MB + event, query/sleep wait", 4}, {"kernel (20 us) x 10 + one event", 5},
        {"kernel (1 ms) + hipEventSynchronize (blocking-sync event)", 6}, {"kernel (1 ms) + hipStreamSynchronize", 7}};
    printf("%-62s %12s %12s %12s\n", "leg", "wall ms/op", "caller us/op", "others us/op");
    for (auto& L : legs) {
        const double w0 = cpu_of(RUSAGE_SELF), m0 = cpu_of(RUSAGE_THREAD);
        timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        for (int i = 0; i < N; i++) {
            switch (L.kind) {
                case 0: case 6: case 7: spin_kernel<<<256, 64, 0, st>>>(d, 400000); break;
                case 1: spin_kernel<<<256, 64, 0, st>>>(d, 8000); break;
                case 2: hipMemcpyAsync(h, d, 512, hipMemcpyDeviceToHost, st); break;
                case 3: hipMemcpyAsync(d, h, 512, hipMemcpyHostToDevice, st); break;
                case 4: hipMemcpyAsync(h, d, 1 << 20, hipMemcpyDeviceToHost, st); break;
                case 5: for (int k = 0; k < 10; k++) spin_kernel<<<256, 64, 0, st>>>(d, 8000); break;
            }
            if (L.kind == 6) { hipEventRecord(evb, st); hipEventSynchronize(evb); }
            else if (L.kind == 7) hipStreamSynchronize(st);
            else { hipEventRecord(ev, st); wait_sleeping(ev); }
        }
        clock_gettime(CLOCK_MONOTONIC, &b);
        const double w1 = cpu_of(RUSAGE_SELF), m1 = cpu_of(RUSAGE_THREAD);
        printf("%-62s %12.3f %12.1f %12.1f\n", L.name, ((b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6) / N, (m1 - m0) * 1e6 / N, ((w1 - w0) - (m1 - m0)) * 1e6 / N);
    }
    return 0;
}
