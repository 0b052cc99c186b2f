// extern "C" wrappers of the native BLS12-381 tower (crate::native in the reference).
#include <stdio.h>

#include "../../include/starkhip.h"
#include "native.h"

using namespace starkhip::bls;

static Fp fp_from(const uint32_t* l) { Fp r; for (int i = 0; i < 12; i++) r.l[i] = l[i]; return r; }
static Fp2 fp2_from(const uint32_t* l) { return Fp2(fp_from(l), fp_from(l + 12)); }

extern "C" {

int starkhip_native_fp12_mul(const uint32_t x[144], const uint32_t y[144], uint32_t out[144]) {
    try {
        (Fp12::from_limbs(x) * Fp12::from_limbs(y)).to_limbs(out);
    } catch (const std::exception& e) { fprintf(stderr, "starkhip native: %s\n", e.what()); return STARKHIP_ERR_BAD_SHAPE; }
    return STARKHIP_OK;
}
int starkhip_native_final_exponentiate(const uint32_t x[144], uint32_t out[144]) {
    try {
        Fp12::from_limbs(x).final_exponentiate().to_limbs(out);
    } catch (const std::exception& e) { fprintf(stderr, "starkhip native: %s\n", e.what()); return STARKHIP_ERR_BAD_SHAPE; }
    return STARKHIP_OK;
}
int starkhip_native_miller_loop(const uint32_t px[12], const uint32_t py[12], const uint32_t qx[24], const uint32_t qy[24], const uint32_t qz[24],
                                uint32_t out[144]) {
    try {
        miller_loop(fp_from(px), fp_from(py), fp2_from(qx), fp2_from(qy), fp2_from(qz)).to_limbs(out);
    } catch (const std::exception& e) { fprintf(stderr, "starkhip native: %s\n", e.what()); return STARKHIP_ERR_BAD_SHAPE; }
    return STARKHIP_OK;
}
int starkhip_native_pairing_precomp(const uint32_t qx[24], const uint32_t qy[24], const uint32_t qz[24], uint32_t out[68 * 72]) {
    try {
        std::vector<EllCoeff> e = calc_pairing_precomp(fp2_from(qx), fp2_from(qy), fp2_from(qz));
        if (e.size() != 68) return STARKHIP_ERR_BAD_SHAPE;
        for (size_t i = 0; i < 68; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 2; k++)
                    for (int l = 0; l < 12; l++) out[i * 72 + j * 24 + k * 12 + l] = e[i][j].c[k].l[l];
    } catch (const std::exception& ex) { fprintf(stderr, "starkhip native: %s\n", ex.what()); return STARKHIP_ERR_BAD_SHAPE; }
    return STARKHIP_OK;
}

}  // extern "C"
