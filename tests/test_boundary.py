"""The drop-in boundary without a GPU: the product library loads, exports every
symbol the headers declare, mirrors the struct layout, refuses to compute
without a device, and its host-only parts (tree utilities, gamma, eigen) agree
with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import common
import pllhip_ctypes as pc
from conftest import ROOT


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set()
    for m in re.finditer(r"PLL_EXPORT\s+(?:extern\s+)?[^;(]*?\b(pll[a-z_0-9]*)\s*(\(|\[|;)", txt):
        names.add(m.group(1))
    return names


def test_exports_every_declared_symbol(product_nogpu):
    declared = _declared("pll.h") | _declared("pllhip.h")
    assert len(declared) > 90
    missing = [n for n in sorted(declared) if not hasattr(product_nogpu.lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    for n in pc.PLL_H_FUNCTIONS + pc.PLLHIP_H_FUNCTIONS:
        assert n in declared, n


def test_oracle_exports_the_same_pll_interface(oracle):
    missing = [n for n in sorted(_declared("pll.h")) if not hasattr(oracle.lib, n)]
    assert not missing


def test_struct_layout_matches_header(tmp_path):
    """sizeof/offsetof of the C structs as the C compiler sees them"""
    src = tmp_path / "layout.c"
    src.write_text('#include "pll.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(pll_partition_t),'
                   'offsetof(pll_partition_t, clv), offsetof(pll_partition_t, tipmap),'
                   'offsetof(pll_partition_t, engine), sizeof(pll_operation_t),'
                   'sizeof(pll_unode_t), sizeof(pll_utree_t)); return 0;}')
    import subprocess
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    assert got == [C.sizeof(pc.Partition), pc.Partition.clv.offset, pc.Partition.tipmap.offset,
                   pc.Partition.engine.offset, C.sizeof(pc.Operation), C.sizeof(pc.UNode),
                   C.sizeof(pc.UTree)]
    assert C.sizeof(pc.Operation) == 32


def test_product_fails_loudly_without_device(product_nogpu):
    if product_nogpu.lib.pllhip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    p = product_nogpu.lib.pll_partition_create(3, 1, 4, 4, 1, 3, 4, 0, 0)
    assert not p
    assert product_nogpu.errno == 901            # PLL_ERROR_HIP_NODEVICE
    assert "no CPU fallback" in product_nogpu.errmsg


def test_gamma_cats_product_equals_oracle(product_nogpu, oracle):
    for alpha in (0.05, 0.3, 0.841, 1.0, 2.5, 20.0, 99.0):
        for k in (2, 4, 8, 16):
            for mode in (0, 1):
                a = product_nogpu.gamma_cats(alpha, k, mode)
                b = oracle.gamma_cats(alpha, k, mode)
                assert np.allclose(a, b, rtol=1e-12, atol=1e-300), (alpha, k, mode)
                assert abs(a.mean() - 1.0) < 1e-5


@pytest.mark.parametrize("states", [2, 4, 5, 20, 61])
def test_eigen_decomposition_reconstructs_rate_matrix(product_nogpu, states):
    """Householder/QL eigen (product) must satisfy V L V^-1 = Q, V^-1 V = I, and give
    the same P(t) as the oracle's Jacobi solver"""
    S = states
    Sp = S if S <= 2 else (S + 3) & ~3
    if S == 20:
        ex, pi = pc.protein_model()
    elif S == 61:
        ex, pi = pc.codon_model()
    elif S == 4:
        ex, pi = np.array(pc.DNA_GTR_RATES), np.array(pc.DNA_FREQS)
    else:
        ex = 0.5 + pc.uniform01(7, S * (S - 1) // 2)
        pi = 0.5 + pc.uniform01(8, S)
        pi /= pi.sum()
    ev = np.zeros(S * Sp); iv = np.zeros(S * Sp); lam = np.zeros(Sp)
    dp = pc.c_double_p
    assert product_nogpu.lib.pllhip_eigen_decompose(
        S, Sp, np.ascontiguousarray(ex).ctypes.data_as(dp), np.ascontiguousarray(pi).ctypes.data_as(dp),
        ev.ctypes.data_as(dp), iv.ctypes.data_as(dp), lam.ctypes.data_as(dp))
    # libpll-2 storage convention: `inv_eigenvecs` holds V, `eigenvecs` holds V^-1
    V = iv.reshape(S, Sp)[:, :S]; Vi = ev.reshape(S, Sp)[:, :S]; L = lam[:S]
    Q = np.zeros((S, S)); k = 0
    for i in range(S):
        for j in range(i + 1, S):
            Q[i, j] = ex[k] * pi[j]; Q[j, i] = ex[k] * pi[i]; k += 1
    Q[np.diag_indices(S)] = -Q.sum(axis=1)
    Q /= -(pi * np.diag(Q)).sum()
    assert np.allclose(V @ np.diag(L) @ Vi, Q, atol=1e-11)
    assert np.allclose(Vi @ V, np.eye(S), atol=1e-11)
    assert abs(L.max()) < 1e-10 and (L <= 1e-10).all()


def _cb_all(node):
    return 1


def test_tree_utilities_host_only(product_nogpu):
    """B2 functions are pure host code: usable without a device.  A full post-order
    traversal of an n-tip tree has 2n-2 records and n-2 operations
    (src/tree/treeinfo.c:984-992), children precede parents."""
    L = product_nogpu.lib
    t = pc.Tree(12)
    tree = L.pll_utree_parse_newick_string(t.newick().encode())
    assert tree, product_nogpu.errmsg
    tr = tree.contents
    assert (tr.tip_count, tr.inner_count, tr.edge_count, tr.binary) == (12, 10, 21, 1)
    assert L.pll_utree_check_integrity(tree)
    buf = (C.POINTER(pc.UNode) * 40)()
    n = C.c_uint()
    cb = pc.TRAVERSE_CB(_cb_all)
    assert L.pll_utree_traverse(tr.vroot, pc.PLL_TREE_TRAVERSE_POSTORDER, cb, buf, C.byref(n))
    assert n.value == 2 * 12 - 2
    ops = (pc.Operation * 12)()
    brl = (C.c_double * 40)(); pmi = (C.c_uint * 40)()
    nm, no = C.c_uint(), C.c_uint()
    L.pll_utree_create_operations(buf, n.value, brl, pmi, ops, C.byref(nm), C.byref(no))
    assert no.value == 10 and nm.value == 21
    assert sorted(pmi[k] for k in range(21)) == list(range(21))
    done = set(range(12))
    for k in range(10):
        assert ops[k].child1_clv_index in done and ops[k].child2_clv_index in done
        done.add(ops[k].parent_clv_index)
        assert ops[k].parent_scaler_index == ops[k].parent_clv_index - 12
    # NULL outputs are legal (src/tree/treeinfo.c:1009-1015)
    L.pll_utree_create_operations(buf, n.value, None, None, ops, None, C.byref(no))
    assert no.value == 10
    # tips cannot be traversal roots
    assert not L.pll_utree_traverse(tr.nodes[0], pc.PLL_TREE_TRAVERSE_POSTORDER, cb, buf, C.byref(n))
    # newick round trip through clone
    clone = L.pll_utree_clone(tree)
    s1 = C.string_at(L.pll_utree_export_newick(tr.vroot, None))
    s2 = C.string_at(L.pll_utree_export_newick(clone.contents.vroot, None))
    assert s1 == s2 and s1.count(b",") == 11
    L.pll_utree_destroy(clone, None)
    L.pll_utree_destroy(tree, None)


def test_partial_traversal_skips_valid_subtrees(product_nogpu):
    """the skip-callback contract used by treeinfo's incremental mode
    (src/tree/treeinfo.c:38-61): cb()==0 drops the node and everything below it"""
    L = product_nogpu.lib
    tree = L.pll_utree_parse_newick_string(b"((a:1,b:1):1,(c:1,d:1):1,(e:1,(f:1,g:1):1):1);")
    tr = tree.contents
    skip = {tr.vroot.contents.next.contents.back.contents.clv_index}

    def cb(node):
        nd = node.contents
        return 0 if (not nd.next) or nd.clv_index in skip else 1
    buf = (C.POINTER(pc.UNode) * 20)(); n = C.c_uint()
    assert L.pll_utree_traverse(tr.vroot, 1, pc.TRAVERSE_CB(cb), buf, C.byref(n))
    got = [buf[k].contents.clv_index for k in range(n.value)]
    assert tr.vroot.contents.clv_index == got[-1]
    assert not (skip & set(got)) and len(got) < 5
    L.pll_utree_destroy(tree, None)
