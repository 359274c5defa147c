// C++ host-layer test (include/sylow_hip.hpp): the shapes of the reference's own tests
// (pairing.rs:1052-1072, lib.rs:29-42) on small batches.  Prints results for the pytest wrapper
// (tests/test_gpu_cpp_host.py), which compares them with the golden fixtures and the oracle.
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "sylow_hip.hpp"

using namespace sylow;

static void print_fp(const Fp& a) { std::printf("%016lx%016lx%016lx%016lx", (unsigned long)a.w[3], (unsigned long)a.w[2], (unsigned long)a.w[1], (unsigned long)a.w[0]); }

int main() {
  try {
    check(sylow_hip_init(0), "sylow_hip_init");
    // test_gt_generator: e(G1gen, G2gen)
    auto gt = pairing({g1_generator()}, {g2_generator()});
    std::printf("GT");
    for (int i = 0; i < 12; ++i) { std::printf(" "); print_fp(gt[0].v.c[i]); }
    std::printf("\n");
    // test_signatures / lib.rs doc-test on 4 keys
    std::vector<Fp> sk = {Fp{{5, 0, 0, 0}}, Fp{{0x1234567890abcdefull, 7, 0, 0}}, Fp{{1, 2, 3, 4}}, Fp{{0xffffffffffffffffull, 0xffffffffffffffffull, 0xffffffffffffffffull, 0x1fffffffffffffffull}}};
    std::vector<std::vector<uint8_t>> msgs(4, std::vector<uint8_t>{0, 0, 0, 20});
    auto sig = sign(sk, msgs);
    auto pk = mul(std::vector<G2Affine>(4, g2_generator()), sk);
    {   // the endomorphism-split product on r-torsion inputs gives the same keys
        auto pk_split = mul(std::vector<G2Affine>(4, g2_generator()), sk, nullptr, nullptr, /*in_subgroup=*/true);
        bool same = true;
        for (int i = 0; i < 4; ++i) same = same && std::memcmp(&pk[i], &pk_split[i], sizeof(G2Affine)) == 0;
        std::printf("G2SPLIT %d\n", same ? 1 : 0);
    }
    auto ok = verify(pk, msgs, sig);
    msgs[2] = {1, 2, 3};
    auto bad = verify(pk, msgs, sig);
    std::printf("VERIFY %d%d%d%d %d%d%d%d\n", ok[0], ok[1], ok[2], ok[3], bad[0], bad[1], bad[2], bad[3]);
    std::printf("SIG0 "); print_fp(sig[0].x); std::printf(" "); print_fp(sig[0].y); std::printf("\n");
    // bilinearity through the glued product: e(5 G1, G2) * e(-G1... ) shape: e(aG1, G2) == e(G1, aG2)
    auto lhs = pairing(mul(std::vector<G1Affine>(1, g1_generator()), {sk[0]}), {g2_generator()});
    auto rhs = pairing({g1_generator()}, {pk[0]});
    std::printf("BILINEAR %d\n", lhs[0] == rhs[0] ? 1 : 0);
    Gt prod = glued_pairing({g1_generator(), g1_generator()}, {g2_generator(), g2_generator()});
    auto sq = pairing({G1Affine{Fp{{1, 0, 0, 0}}, Fp{{2, 0, 0, 0}}}}, {mul(std::vector<G2Affine>(1, g2_generator()), {Fp{{2, 0, 0, 0}}})[0]});
    std::printf("GLUED %d\n", prod == sq[0] ? 1 : 0);
    // gt::tests::test_bilinearity shape: e(P, Q) * s == e(s P, Q); Fr: s * s^-1 == 1; aggregate: 2 P + 3 P == 5 P
    auto es = mul(rhs, {sk[0]});                              // e(G1, 5 G2) * 5
    auto e25 = pairing(mul(std::vector<G1Affine>(1, g1_generator()), {Fp{{25, 0, 0, 0}}}), {g2_generator()});
    std::printf("GTPOW %d\n", es[0] == e25[0] ? 1 : 0);
    auto one = fr::mul({sk[1]}, fr::inv({sk[1]}));
    std::printf("FRINV %d\n", (one[0].w[0] == 1 && !one[0].w[1] && !one[0].w[2] && !one[0].w[3]) ? 1 : 0);
    auto agg = aggregate({g1_generator(), g1_generator()}, {Fp{{2, 0, 0, 0}}, Fp{{3, 0, 0, 0}}}, 1, 2);
    auto five = mul(std::vector<G1Affine>(1, g1_generator()), {sk[0]});
    bool same = true;
    for (int i = 0; i < 4; ++i) same = same && agg[0].x.w[i] == five[0].x.w[i] && agg[0].y.w[i] == five[0].y.w[i];
    std::printf("AGG %d\n", same ? 1 : 0);
    // identity handling through the flags (pairing.rs:876-886: an identity operand maps to Gt::identity()):
    //  * sk = 0 signs to the identity and the flag comes back; 0 * G2gen is the identity key
    //  * verify with BOTH flags set: identity == identity -> true; with only one side the identity -> false
    //  * the same coordinates WITHOUT the flags are just an off-curve pair and must not verify
    std::vector<Fp> sk0 = {Fp{{0, 0, 0, 0}}, sk[0]};
    std::vector<std::vector<uint8_t>> m2(2, std::vector<uint8_t>{9, 9});
    std::vector<uint8_t> sinf, pinf;
    auto sig2 = sign(sk0, m2, &sinf);
    auto pk2 = mul(std::vector<G2Affine>(2, g2_generator()), sk0, &pinf);
    auto v_flags = verify(pk2, m2, sig2, &pinf, &sinf);
    std::vector<uint8_t> only_sig = {1, 0}, none = {0, 0};
    auto v_half = verify(pk2, m2, sig2, &none, &only_sig);
    auto v_noflag = verify(pk2, m2, sig2);
    std::printf("IDENT %d%d %d%d %d%d %d%d %d%d\n", sinf[0], sinf[1], pinf[0], pinf[1], v_flags[0], v_flags[1], v_half[0], v_half[1], v_noflag[0], v_noflag[1]);
    // glued product with an identity G1 flagged and skipped (EIP-197 semantics) == the product of the rest
    std::vector<uint8_t> gi = {1, 0}, qi = {0, 0};
    Gt skipped = glued_pairing({g1_generator(), g1_generator()}, {g2_generator(), g2_generator()}, &gi, &qi, true);
    std::printf("GLUEDSKIP %d\n", skipped == gt[0] ? 1 : 0);
    // one boolean for a batch: unweighted product and the weighted (sound) test; P - P is the identity
    std::vector<Fp> sk3 = {sk[0], Fp{{7, 0, 0, 0}}, Fp{{11, 0, 0, 0}}};
    std::vector<std::vector<uint8_t>> m3 = {{1}, {2, 2}, {3, 3, 3}};
    auto sig3 = sign(sk3, m3);
    auto pk3 = mul(std::vector<G2Affine>(3, g2_generator()), sk3, nullptr, nullptr, true);
    std::vector<Fp> w3 = {Fp{{0x9e3779b97f4a7c15ull, 1, 0, 0}}, Fp{{0xbf58476d1ce4e5b9ull, 2, 0, 0}}, Fp{{0x94d049bb133111ebull, 3, 0, 0}}};
    const bool all_plain = verify_all(pk3, m3, sig3), all_weighted = verify_all(pk3, m3, sig3, &w3);
    std::swap(sig3[0], sig3[1]);
    const bool bad_weighted = verify_all(pk3, m3, sig3, &w3);
    std::vector<uint8_t> dinf;
    auto diff = sub({g1_generator()}, {g1_generator()}, &dinf);
    std::printf("VERIFYALL %d%d%d SUB %d\n", all_plain ? 1 : 0, all_weighted ? 1 : 0, bad_weighted ? 1 : 0, dinf[0]);
    // host vectors in / out through the chunked two-stream pipeline (several chunks: 70000 > 2^16) == upload -> batch call -> download
    {
      const size_t n = 70000;
      std::vector<Fp> ks(n);
      for (size_t i = 0; i < n; ++i) ks[i] = Fp{{0x9e3779b97f4a7c15ull * (i + 1), i, 7, 0}};
      auto ps = mul(std::vector<G1Affine>(n, g1_generator()), ks);
      auto qs = mul(std::vector<G2Affine>(n, g2_generator()), ks, nullptr, nullptr, true);
      std::vector<uint8_t> pi(n, 0), qi(n, 0);
      pi[3] = 1; qi[n - 1] = 1; pi[65536] = 1;
      const bool same = pairing(ps, qs, &pi, &qi) == pairing_unpipelined(ps, qs, &pi, &qi);
      std::vector<std::vector<uint8_t>> ms(n);
      for (size_t i = 0; i < n; ++i) ms[i].assign(i % 37, (uint8_t)i);
      auto sg = sign(ks, ms);
      std::swap(sg[10], sg[11]);
      auto v1 = verify(qs, ms, sg), v2 = verify_unpipelined(qs, ms, sg);
      size_t good = 0;
      for (auto f : v1) good += f;
      // page-locked staging through the allocator
      std::vector<G1Affine, PinnedAllocator<G1Affine>> pp(ps.begin(), ps.end());
      std::vector<G2Affine, PinnedAllocator<G2Affine>> qp(qs.begin(), qs.end());
      std::vector<Gt, PinnedAllocator<Gt>> gp(n);
      check(sylow_hip_pairing_host(reinterpret_cast<const uint64_t*>(pp.data()), nullptr, reinterpret_cast<const uint64_t*>(qp.data()), nullptr,
                                   reinterpret_cast<uint64_t*>(gp.data()), n, 0), "pairing_host(pinned)");
      auto plain = pairing(ps, qs);
      bool pinned_same = true;
      for (size_t i = 0; i < n; ++i) pinned_same = pinned_same && (gp[i] == plain[i]);
      std::printf("PIPELINE %d%d%d %zu\n", same ? 1 : 0, v1 == v2 ? 1 : 0, pinned_same ? 1 : 0, n - good);
      // round 6: route options and the live clock probe through the C++ layer.  The same 70000 pairings with the quad tail switched off
      // (TAIL_SPLIT 0: 70000 = two rounds of 32768 + 4464) and with the skew muted; the probe armed around a skewed launch of 2^17 + 5.
      set_option(SYLOW_HIP_OPT_TAIL_SPLIT, 0);
      auto no_tail = pairing(ps, qs);
      set_option(SYLOW_HIP_OPT_TAIL_SPLIT, -1);
      bool opt_same = get_option(SYLOW_HIP_OPT_TAIL_SPLIT) == -1;
      for (size_t i = 0; i < n; ++i) opt_same = opt_same && (no_tail[i] == plain[i]);
      std::vector<G1Affine> pb((1u << 17) + 5, ps[1]);
      std::vector<G2Affine> qb(pb.size(), qs[1]);
      ClockProbe probe;
      probe.arm();
      auto big = pairing(pb, qb);
      uint64_t waves = 0;
      const double mhz = probe.read_mhz(&waves);
      bool big_same = true;
      for (size_t i = 0; i < big.size(); ++i) big_same = big_same && (big[i] == plain[1]);
      std::printf("OPTIONS %d%d %d %d\n", opt_same ? 1 : 0, big_same ? 1 : 0, (mhz > 500.0 && mhz < 3500.0) ? 1 : 0, (waves >= 4000 && waves <= 4 * ((2 * pb.size() + 255) / 256 + 512)) ? 1 : 0);   // the lane-pair launches' wavefronts (a quad tail carries no probe; a skewed launch adds finishing blocks)
    }
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
}
