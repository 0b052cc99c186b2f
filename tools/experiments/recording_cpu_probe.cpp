// CPU time, kernel time and page faults of one FinalExp recording on the host (round 6: the 64-bit division, the row spans and the recycled log
// vectors were measured with this).  g++ -O2 -std=c++17 -Iinclude -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/experiments/recording_cpu_probe.cpp \
//   starky_bls12_381_amd/csrc/*.cpp starky_bls12_381_amd/csrc/host_only_stubs.cc -lpthread -o build/recording_cpu_probe ;  usage: recording_cpu_probe [recordings] [threads]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/resource.h>
#include <vector>
#include "starkhip.h"
int main(int argc, char** argv) {
    uint32_t x[144];
    for (int i = 0; i < 144; i++) x[i] = (i % 12 == 11) ? 0x08000000u : 0x5000u + 3u * i;
    starkhip_trace_set_threads(argc > 2 ? atoi(argv[2]) : 1);
    int reps = argc > 1 ? atoi(argv[1]) : 5;
    for (int r = 0; r < reps; r++) {
        rusage a, b; getrusage(RUSAGE_SELF, &a);
        void* log = nullptr;
        starkhip_trace_log_begin(&log);
        std::vector<uint64_t> pis(6000);
        int rc = starkhip_trace_final_exp(x, nullptr, 8192, pis.data());
        starkhip_trace_log_end(log);
        size_t rows, cols, recs, words; starkhip_trace_log_info(log, &rows, &cols, &recs, &words);
        starkhip_trace_log_free(log);
        getrusage(RUSAGE_SELF, &b);
        auto tv=[](timeval t){return t.tv_sec+1e-6*t.tv_usec;};
        printf("rc %d user %.3f sys %.3f minflt %ld words %zu recs %zu\n", rc, tv(b.ru_utime)-tv(a.ru_utime), tv(b.ru_stime)-tv(a.ru_stime), b.ru_minflt-a.ru_minflt, words, recs);
    }
}
