// Host-side Poseidon permutation used by the Fiat-Shamir challenger and the verifier.  The challenger absorbs
// 2 * (2C + Q) field elements per proof (36.8 K sequential permutations for FinalExp), so this sits on the proof's
// critical path between the openings and the FRI combination.  Same permutation as poseidon.h; the MDS layer is
// arranged so the compiler can use 32x32->64 vector multiplies; an AVX2 build of the same body is picked at run time.
#include "poseidon.h"

namespace starkhip {

static inline __attribute__((always_inline)) void permute_body(gl_t* s) {
    static const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint64_t* RC = POSEIDON_RC_HOST;
    for (int round = 0; round < 30; round++) {
        const uint64_t* rc = RC + 12 * round;
        if (round < 4 || round >= 26) {
            for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(gl_add(s[i], rc[i]));
        } else {
            for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], rc[i]);
            s[0] = poseidon_sbox(s[0]);
        }
        alignas(32) uint32_t lo[24], hi[24];
        for (int i = 0; i < 12; i++) {
            lo[i] = lo[i + 12] = (uint32_t)s[i];
            hi[i] = hi[i + 12] = (uint32_t)(s[i] >> 32);
        }
        alignas(32) uint64_t al[12] = {0}, ah[12] = {0};
        for (int i = 0; i < 12; i++) {
            const uint64_t c = CIRC[i];
            for (int r = 0; r < 12; r++) {
                al[r] += (uint64_t)lo[i + r] * c;
                ah[r] += (uint64_t)hi[i + r] * c;
            }
        }
        al[0] += (uint64_t)lo[0] * 8;
        ah[0] += (uint64_t)hi[0] * 8;
        for (int r = 0; r < 12; r++) {
            const uint64_t l = al[r] + (ah[r] << 32);
            const uint64_t h = (ah[r] >> 32) + (l < al[r] ? 1 : 0);
            s[r] = gl_reduce128(h, l);
        }
    }
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static void permute_avx2(gl_t* s) { permute_body(s); }
#endif
static void permute_base(gl_t* s) { permute_body(s); }

void poseidon_permute_host(gl_t* s) {
#if defined(__x86_64__)
    static const bool has_avx2 = __builtin_cpu_supports("avx2");
    if (has_avx2) {
        permute_avx2(s);
        return;
    }
#endif
    permute_base(s);
}

}  // namespace starkhip
