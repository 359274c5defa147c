"""The digit masks of the constant-exponent chain shared by f^x (bn254_pair29.hpp: exp_by_neg_z29) and x Q (plk_group.hip:
g2q_in_subgroup_proj; g2.rs:488-513 multiplies by the same x): sum d_i 2^i == x with digits in {0, +-17, +-35}, top digit +35 at bit 57."""
import os
import re

X = 4965661367192848881


def test_expx_chain_masks_encode_x():
    """exp_by_neg_z29 (bn254_pair29.hpp): f^x by a signed-digit chain over {f^17, f^35} -- the three masks and the top digit 35 at bit 57
    reproduce the BN parameter, with 11 products in the loop."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "sylow_amd", "csrc", "bn254_pair29.hpp")).read()
    vals = {}
    for name in ("BN_X_C_NZ", "BN_X_C_NEG", "BN_X_C_17"):
        m = re.search(r"#define %s (0x[0-9a-f]+)ull" % name, src)
        assert m, name
        vals[name] = int(m.group(1), 16)
    nz, neg, is17 = vals["BN_X_C_NZ"], vals["BN_X_C_NEG"], vals["BN_X_C_17"]
    assert neg & ~nz == 0 and is17 & ~nz == 0 and nz >> 57 == 0
    total = 35 << 57
    for i in range(57):
        if (nz >> i) & 1:
            d = 17 if (is17 >> i) & 1 else 35
            total += (-d if (neg >> i) & 1 else d) << i
    assert total == X
    assert bin(nz).count("1") == 11
    assert "for (int i = 56; i >= 0; --i)" in src and "W12 res = tab[1];" in src
    grp = open(os.path.join(os.path.dirname(__file__), "..", "sylow_amd", "csrc", "plk_group.hip")).read()
    assert "constexpr u64 NZ = BN_X_C_NZ, NEG = BN_X_C_NEG, IS17 = BN_X_C_17;" in grp and "G2Q a = q35;" in grp
