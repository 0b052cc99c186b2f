// Toy AIR used by the unit tests of the prover pipeline (not one of the reference's AIRs).
// Columns: x0, x1 (Fibonacci pair), prod = x0*x1, b = row parity.  Public inputs: x0[0], x1[0], x1[n-1].
// Exercises every constraint kind, plain/complement gates, constants and public inputs.
#include "airs.h"

namespace starkhip {

AirProgram build_air_fibonacci() {
    AirBuilder b(4, 3, 3);
    auto one = AirBuilder::one();
    b.first_row(b.L(0) - b.PI(0));
    b.first_row(b.L(1) - b.PI(1));
    b.last_row(b.L(1) - b.PI(2));
    b.transition(b.N(0) - b.L(1));
    b.transition(b.N(1) - b.L(0) - b.L(1));
    b.constraint(b.L(2) - b.L(0) * b.L(1));
    b.constraint(b.L(3) * (one - b.L(3)));
    b.constraint(b.L(3) * (b.L(2) - b.L(0) * b.L(1)) * AirBuilder::C(5));
    b.transition((one - b.L(3)) * (b.N(3) - 1));
    b.transition(b.L(3) * b.N(3));
    b.transition((one - b.L(3)) * (b.L(2) * AirBuilder::C(1ULL << 32) - b.L(0) * b.L(1) * AirBuilder::C(1ULL << 32)));
    return b.finish();
}

}  // namespace starkhip

extern "C" int starkhip_trace_fibonacci(uint64_t x0, uint64_t x1, uint64_t* trace, size_t n_rows, uint64_t* public_inputs) {
    if (n_rows < 2 || (n_rows & (n_rows - 1))) return STARKHIP_ERR_BAD_SHAPE;
    gl_t a = gl_from_u64(x0), c = gl_from_u64(x1);
    public_inputs[0] = a;
    public_inputs[1] = c;
    for (size_t i = 0; i < n_rows; i++) {
        trace[4 * i + 0] = a;
        trace[4 * i + 1] = c;
        trace[4 * i + 2] = gl_mul(a, c);
        trace[4 * i + 3] = i & 1;
        public_inputs[2] = c;
        gl_t t = gl_add(a, c);
        a = c;
        c = t;
    }
    return STARKHIP_OK;
}
