// CPU verifier: the counterpart of starky::verifier::verify_stark_proof, which the reference calls
// right after every prove (/root/reference/src/aggregate_proof.rs:67,113,146,177).  Written from the
// verifier's equations, i.e. an independent code path from the prover: it
// re-derives every Fiat-Shamir challenge, checks the quotient identity at zeta with an extension-field
// evaluation of the AIR, and runs the FRI query checks.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "air_eval.h"
#include "airs.h"
#include "poseidon.h"
#include "proof.h"

namespace starkhip {

static bool merkle_verify_to_cap(const gl_t* leaf, size_t leaf_len, size_t index, const gl_t* cap, const gl_t* siblings, size_t depth) {
    gl_t cur[4];
    poseidon_hash_or_noop(leaf, leaf_len, 1, cur);
    for (size_t d = 0; d < depth; d++) {
        gl_t nxt[4];
        if (index & 1) poseidon_two_to_one(siblings + 4 * d, cur, nxt);
        else poseidon_two_to_one(cur, siblings + 4 * d, nxt);
        memcpy(cur, nxt, sizeof cur);
        index >>= 1;
    }
    return memcmp(cur, cap + 4 * index, sizeof cur) == 0;
}

static gl2_t eval_poly_ext(const gl2_t* coeffs, size_t n, gl2_t x) {
    gl2_t acc = gl2_zero();
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, x), coeffs[i]);
    return acc;
}

// plonky2 fri::verifier::compute_evaluation: interpolate the arity coset values and evaluate at beta.
static gl2_t fri_fold_eval(gl_t x, size_t x_index_within_coset, unsigned arity_bits, const gl2_t* evals_in, gl2_t beta) {
    size_t arity = (size_t)1 << arity_bits;
    gl_t g = gl_root_of_unity(arity_bits);
    std::vector<gl2_t> evals(arity);
    for (size_t i = 0; i < arity; i++) evals[gl_bitrev((uint32_t)i, arity_bits)] = evals_in[i];
    size_t rev = gl_bitrev((uint32_t)x_index_within_coset, arity_bits);
    gl_t coset_start = gl_mul(x, gl_pow(g, arity - rev));
    std::vector<gl_t> pts(arity);
    gl_t y = 1;
    for (size_t i = 0; i < arity; i++) {
        pts[i] = gl_mul(coset_start, y);
        y = gl_mul(y, g);
    }
    // Lagrange interpolation at beta
    gl2_t res = gl2_zero();
    for (size_t i = 0; i < arity; i++) {
        gl2_t num = gl2_one();
        gl_t den = 1;
        for (size_t j = 0; j < arity; j++) {
            if (j == i) continue;
            num = gl2_mul(num, gl2_sub(beta, gl2_from_base(pts[j])));
            den = gl_mul(den, gl_sub(pts[i], pts[j]));
        }
        res = gl2_add(res, gl2_mul(evals[i], gl2_mul_base(num, gl_inv(den))));
    }
    return res;
}

int verify_proof(const AirInfo& air, const starkhip_config_t& cfg, const uint64_t* proof, size_t words) {
    ProofLayout pl;
    if (!pl.read_header(proof, words)) return STARKHIP_ERR_BAD_SHAPE;
    const AirProgram& P = air.prog;
    const unsigned factor = P.degree > 1 ? P.degree - 1 : 1;
    if (pl.C != P.n_cols || pl.n_pis != P.n_pis || pl.rate_bits != cfg.rate_bits || pl.cap_h != cfg.cap_height ||
        pl.n_queries != cfg.num_query_rounds || pl.n_challenges != cfg.num_challenges || pl.Q != (size_t)factor * cfg.num_challenges ||
        pl.arity_bits != cfg.arity_bits)
        return STARKHIP_ERR_BAD_SHAPE;
    for (size_t i = 16; i < words; i++)
        if (proof[i] >= GL_P) return STARKHIP_ERR_BAD_SHAPE;
    FriGeometry geo;
    if (!FriGeometry::make(cfg, (unsigned)pl.log_n, &geo)) return STARKHIP_ERR_BAD_SHAPE;
    if (geo.arities.size() != pl.L || geo.final_poly_len != pl.final_len) return STARKHIP_ERR_BAD_SHAPE;

    const size_t C = pl.C, Q = pl.Q, n = (size_t)1 << pl.log_n, N = (size_t)1 << pl.log_N, ncap = pl.ncap;
    const gl2_t* op_local = (const gl2_t*)(proof + pl.off_local);
    const gl2_t* op_next = (const gl2_t*)(proof + pl.off_next);
    const gl2_t* op_q = (const gl2_t*)(proof + pl.off_quot_open);
    const gl2_t* final_poly = (const gl2_t*)(proof + pl.off_final);
    const gl_t* pis = proof + pl.off_pis;

    // ---- challenges, in transcript order (App. A.5)
    Challenger ch;
    ch.observe_many(proof + pl.off_trace_cap, 4 * ncap);
    std::vector<gl2_t> alphas(cfg.num_challenges);
    for (auto& a : alphas) a = gl2_from_base(ch.get());
    ch.observe_many(proof + pl.off_quot_cap, 4 * ncap);
    gl2_t zeta = ch.get_ext();
    for (size_t c = 0; c < C; c++) ch.observe_ext(op_local[c]);
    for (size_t q = 0; q < Q; q++) ch.observe_ext(op_q[q]);
    for (size_t c = 0; c < C; c++) ch.observe_ext(op_next[c]);
    gl2_t fri_alpha = ch.get_ext();
    std::vector<gl2_t> betas(pl.L);
    for (size_t l = 0; l < pl.L; l++) {
        ch.observe_many(proof + pl.off_fri_caps + l * 4 * ncap, 4 * ncap);
        betas[l] = ch.get_ext();
    }
    for (size_t k = 0; k < pl.final_len; k++) ch.observe_ext(final_poly[k]);
    ch.observe(proof[pl.off_pow]);
    gl_t pow_response = ch.get();
    if (cfg.proof_of_work_bits > 0 && (pow_response >> (64 - cfg.proof_of_work_bits)) != 0) return STARKHIP_ERR_VERIFY;
    std::vector<size_t> indices(pl.n_queries);
    for (auto& x : indices) x = (size_t)(ch.get() % N);

    // ---- quotient identity at zeta
    gl_t g = gl_root_of_unity((unsigned)pl.log_n);
    gl2_t zeta_n = gl2_pow(zeta, n);
    gl2_t z_h = gl2_sub(zeta_n, gl2_one());
    gl2_t masks[4];
    masks[KIND_PLAIN] = gl2_one();
    masks[KIND_TRANSITION] = gl2_sub(zeta, gl2_from_base(gl_inv(g)));
    masks[KIND_FIRST] = gl2_mul(z_h, gl2_inv(gl2_mul_base(gl2_sub(zeta, gl2_one()), (gl_t)n)));
    masks[KIND_LAST] = gl2_mul(z_h, gl2_inv(gl2_mul_base(gl2_sub(gl2_mul_base(zeta, g), gl2_one()), (gl_t)n)));
    std::vector<gl2_t> acc(cfg.num_challenges);
    air_eval_folded<ExtOps>(P, op_local, op_next, pis, masks, alphas.data(), (int)cfg.num_challenges, acc.data());
    for (unsigned i = 0; i < cfg.num_challenges; i++) {
        gl2_t s = gl2_zero();
        for (unsigned k = factor; k-- > 0;) s = gl2_add(gl2_mul(s, zeta_n), op_q[i * factor + k]);
        if (!gl2_eq(acc[i], gl2_mul(z_h, s))) return STARKHIP_ERR_VERIFY;
    }

    // ---- FRI
    gl2_t gzeta = gl2_mul_base(zeta, g);
    // precomputed reduced openings per batch: sum_j alpha^j open_j
    gl2_t red0 = gl2_zero(), red1 = gl2_zero();
    for (size_t q = Q; q-- > 0;) red0 = gl2_add(gl2_mul(red0, fri_alpha), op_q[q]);
    for (size_t c = C; c-- > 0;) red0 = gl2_add(gl2_mul(red0, fri_alpha), op_local[c]);
    for (size_t c = C; c-- > 0;) red1 = gl2_add(gl2_mul(red1, fri_alpha), op_next[c]);
    gl2_t alpha_pow_C = gl2_pow(fri_alpha, C);

    // The 84 query rounds are independent and dominated by re-hashing a trace leaf each (FinalExp: 9191 permutations per
    // leaf, 0.8 M in all -- about a second on one core), so they are checked by a few host threads.
    const size_t d0 = pl.log_N - pl.cap_h;
    auto check_query = [&](size_t qi) -> bool {
        const uint64_t* qp = proof + pl.off_queries + qi * pl.query_words;
        size_t x_index = indices[qi];
        const gl_t* tleaf = qp; qp += C;
        const gl_t* tsib = qp; qp += 4 * d0;
        const gl_t* qleaf = qp; qp += Q;
        const gl_t* qsib = qp; qp += 4 * d0;
        if (!merkle_verify_to_cap(tleaf, C, x_index, proof + pl.off_trace_cap, tsib, d0)) return false;
        if (!merkle_verify_to_cap(qleaf, Q, x_index, proof + pl.off_quot_cap, qsib, d0)) return false;
        gl_t subgroup_x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity((unsigned)pl.log_N), gl_bitrev((uint32_t)x_index, (unsigned)pl.log_N)));
        // fri_combine_initial
        gl2_t e0 = gl2_zero(), e1 = gl2_zero();
        for (size_t q = Q; q-- > 0;) e0 = gl2_add(gl2_mul(e0, fri_alpha), gl2_from_base(qleaf[q]));
        for (size_t c = C; c-- > 0;) e0 = gl2_add(gl2_mul(e0, fri_alpha), gl2_from_base(tleaf[c]));
        for (size_t c = C; c-- > 0;) e1 = gl2_add(gl2_mul(e1, fri_alpha), gl2_from_base(tleaf[c]));
        gl2_t xe = gl2_from_base(subgroup_x);
        gl2_t sum = gl2_mul(gl2_sub(e0, red0), gl2_inv(gl2_sub(xe, zeta)));
        sum = gl2_add(gl2_mul(sum, alpha_pow_C), gl2_mul(gl2_sub(e1, red1), gl2_inv(gl2_sub(xe, gzeta))));
        gl2_t old_eval = sum;
        for (size_t l = 0; l < pl.L; l++) {
            unsigned ab = geo.arities[l];
            size_t arity = (size_t)1 << ab;
            const gl2_t* evals = (const gl2_t*)qp; qp += 2 * arity;
            const gl_t* sib = qp; qp += 4 * pl.layer_depth[l];
            size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!gl2_eq(evals[within], old_eval)) return false;
            old_eval = fri_fold_eval(subgroup_x, within, ab, evals, betas[l]);
            if (!merkle_verify_to_cap((const gl_t*)evals, 2 * arity, coset_index, proof + pl.off_fri_caps + l * 4 * ncap, sib, pl.layer_depth[l]))
                return false;
            for (unsigned b = 0; b < ab; b++) subgroup_x = gl_sqr(subgroup_x);
            x_index = coset_index;
        }
        if (!gl2_eq(eval_poly_ext(final_poly, pl.final_len, gl2_from_base(subgroup_x)), old_eval)) return false;
        return true;
    };
    const size_t n_threads = std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), pl.n_queries);
    std::atomic<bool> ok(true);
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        for (size_t qi; ok.load(std::memory_order_relaxed) && (qi = next.fetch_add(1)) < pl.n_queries;)
            if (!check_query(qi)) ok.store(false);
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_threads; t++) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    if (!ok.load()) return STARKHIP_ERR_VERIFY;
    return STARKHIP_OK;
}

}  // namespace starkhip
