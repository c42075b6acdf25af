// Host build of plonk_amd/csrc/fp29.cuh and g1.cuh for CPU-side unit tests: the same source the
// GPU kernels compile, driven through a tiny C ABI.  Values cross this ABI as eight 32-bit words.
#include "../../plonk_amd/csrc/g1.cuh"
using namespace plk;

template <class F> static void op(int what, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    fe x = unpack(a), y = b ? unpack(b) : fe_zero(), r;
    switch (what) {
        case 0: r = mul<F>(x, y); break;
        case 1: r = sqr<F>(x); break;
        case 2: r = mul<F>(add(x, y), fe_const(F::ONE)); break;                       // (x+y) mod p
        case 3: { fe d = sub2<F>(x, y); norm(d); r = mul<F>(d, fe_const(F::ONE)); break; }   // (x-y) mod p
        case 4: r = inv<F>(x); break;
        case 5: { fe d = sub2<F>(x, y); norm(d); r = fe_zero(); r.v[0] = is_zero_mod<F>(d) ? 1 : 0; break; }
        case 6: { fe d = sub32<F>(x, y); norm(d); r = fe_zero(); r.v[0] = is_zero_mod<F>(d) ? 1 : 0; break; }
        case 8: { fe t = sub8<F>(x, y); fe n = sub4<F>(fe_zero(), y); r = mul2<F>(x, t, n, y); break; }   // x (x - y) - y y, operands as fat as xyzz_madd makes them
        case 9: {   // Shoup product of a fat x (x + 8p + ..., not normalized) with the fixed operand y: y arrives as y * 2^261 mod p in b
            fe fat = sub8<F>(x, fe_zero());          // x + 8p, limbs up to 2^29 + 2^30
            fat = add(fat, fat);                      // 2x + 16p: limbs up to 2.5 * 2^30, bound 18
            const shoup c = shoup_of<F>(reduce<F>(mul<F>(y, fe_const(F::R2))));
            r = reduce3<F>(mul_shoup<F>(fat, c));     // 2 x y mod p
            break;
        }
        case 10: {  // the same with the negated operand
            const shoup c = shoup_neg<F>(shoup_of<F>(reduce<F>(mul<F>(y, fe_const(F::R2)))));
            r = reduce3<F>(mul_shoup<F>(x, c));       // -x y mod p
            break;
        }
        default: r = x;
    }
    pack(reduce<F>(r), out);
}
static void out_xyzz(const xyzz& p, uint32_t* out) {
    const fe* c[4] = {&p.x, &p.y, &p.zz, &p.zzz};
    for (int i = 0; i < 4; ++i) pack(reduce<FqP>(mul<FqP>(*c[i], fe_const(FqP::TO_R))), out + 8 * i);
}
static aff in_aff(const uint32_t* xy) {   // x*2^256 words -> canonical x*2^261 limbs
    aff q;
    q.x = reduce<FqP>(mul<FqP>(unpack(xy), fe_const(FqP::TO_RP)));
    q.y = reduce<FqP>(mul<FqP>(unpack(xy + 8), fe_const(FqP::TO_RP)));
    return q;
}
extern "C" {
void fp29_fr(int what, const uint32_t* a, const uint32_t* b, uint32_t* out) { op<FrP>(what, a, b, out); }
void fp29_fq(int what, const uint32_t* a, const uint32_t* b, uint32_t* out) { op<FqP>(what, a, b, out); }
// sum of n affine points (x*2^256 words, 16 words each) by repeated madd, optionally negating odd entries;
// returns X, Y, ZZ, ZZZ as x*2^256 words (32 words)
void g1_madd_chain(const uint32_t* pts, int n, int neg_odd, uint32_t* out) {
    xyzz acc = xyzz_inf();
    for (int i = 0; i < n; ++i) { aff q = in_aff(pts + 16 * i); if (neg_odd && (i & 1)) q.y = fq_neg_y(q.y); xyzz_madd(acc, q); }
    out_xyzz(acc, out);
}
// tree: madd pairs into XYZZ partial sums, then xyzz_add them all, then double k times
void g1_add_tree(const uint32_t* pts, int n, int doublings, uint32_t* out) {
    xyzz total = xyzz_inf();
    for (int i = 0; i < n; i += 2) {
        xyzz part = xyzz_inf();
        xyzz_madd(part, in_aff(pts + 16 * i));
        if (i + 1 < n) xyzz_madd(part, in_aff(pts + 16 * (i + 1)));
        total = xyzz_add(total, part);
    }
    for (int k = 0; k < doublings; ++k) total = xyzz_dbl(total);
    out_xyzz(total, out);
}
}
