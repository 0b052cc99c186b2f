// Host-side Poseidon permutation used by the Fiat-Shamir challenger and the verifier.  The challenger absorbs
// 2 * (2C + Q) field elements per proof (36.8 K sequential permutations for FinalExp), so this sits on the proof's
// critical path between the openings and the FRI combination.  Measured on the GPU box's EPYC 9575F
// (tools/host_perm_rate.py): the plain loop with branch-free field ops runs 2.08 us / permutation; an AVX2
// auto-vectorised MDS variant was slower (2.74 us) and was dropped.
#include "poseidon.h"

namespace starkhip {

void poseidon_permute_host(gl_t* s) { poseidon_permute(s); }

}  // namespace starkhip
