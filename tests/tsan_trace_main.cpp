// ThreadSanitizer harness for the threaded trace recording (csrc/trace_tasks.cpp): `make tsan-test` builds the host code with
// -fsanitize=thread and records a MillerLoop, a PairingPrecomp and a FinalExp trace on 6 threads.  Test infrastructure only.
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "starkhip.h"

int main() {
    uint32_t px[12], py[12], q[3][24], x[144];
    for (int i = 0; i < 12; i++) { px[i] = 0x1000u + 7u * i; py[i] = 0x2000u + 11u * i; }
    px[11] = py[11] = 0x0a000000u;  // below the modulus' top limb
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 24; i++) q[k][i] = (i % 12 == 11) ? 0x09000000u : 0x3000u * (k + 1) + 13u * i;
    for (int i = 0; i < 144; i++) x[i] = (i % 12 == 11) ? 0x08000000u : 0x5000u + 3u * i;
    starkhip_trace_set_threads(6);
    int bad = 0;
    for (int which = 0; which < 3; which++) {
        void* log = nullptr;
        if (starkhip_trace_log_begin(&log) != STARKHIP_OK) return 2;
        std::vector<uint64_t> pis(6000);
        int rc = which == 0   ? starkhip_trace_miller_loop(px, py, q[0], q[1], q[2], nullptr, 1024, pis.data())
                 : which == 1 ? starkhip_trace_pairing_precomp(q[0], q[1], q[2], nullptr, 1024, pis.data())
                              : starkhip_trace_final_exp(x, nullptr, 8192, pis.data());
        if (starkhip_trace_log_end(log) != STARKHIP_OK || rc != STARKHIP_OK) bad++;
        size_t rows, cols, recs, words;
        starkhip_trace_log_info(log, &rows, &cols, &recs, &words);
        printf("trace %d: rc %d, %zu x %zu, %zu records, %zu words\n", which, rc, rows, cols, recs, words);
        starkhip_trace_log_free(log);
    }
    return bad ? 1 : 0;
}
