"""CPU: the host-side readers of libFM's data formats against files written by the REFERENCE's own convert tool
(oracle/_ref/convert, compiled from /root/reference/src/libfm/tools/convert.cpp when the reference is present) and
against the documented sizes (.x = 24 + 4N + 8 nnz bytes, .y = 12 + 4N bytes)."""
import os
import subprocess

import numpy as np
import pytest

import datagen
from conftest import ROOT

CONVERT = os.path.join(ROOT, "oracle", "_ref", "convert")


def test_binary_round_trip_and_sizes(tmp_path):
    from libfm_amd import data as D
    ent, rp, y = datagen.ragged_real(97, 60, 9, seed=3, empty_every=7)
    pre = str(tmp_path / "t")
    D.write_binary(pre, ent, rp, y)
    assert os.path.getsize(pre + ".x") == 24 + 4 * 60 + 8 * len(ent)
    assert os.path.getsize(pre + ".y") == 12 + 4 * 60
    e2, r2, y2 = D.load(pre)
    assert np.array_equal(e2, ent) and np.array_equal(r2, rp) and np.array_equal(y2, y)


def test_text_reader_matches_reference_parser_rules(tmp_path, oracle):
    from libfm_amd import data as D
    p = str(tmp_path / "a.libfm")
    with open(p, "w") as f:
        f.write("# comment\n\n  1 0:1 5:0.5\n\t-1 3:2\n0\n2.5 1:1 2:1 # trailing comment\n")
    ent, rp, y = D.read_libsvm(p)
    assert list(y) == [1.0, -1.0, 0.0, 2.5]
    assert list(np.diff(rp.astype(int))) == [2, 1, 0, 2]
    assert list(ent["id"]) == [0, 5, 3, 1, 2]


def test_native_text_reader_against_python_and_reference_errors(tmp_path, oracle):
    """fmx_read_libsvm (C-ABI, host side) == the pure-Python reader on a seeded file, and it rejects what Data::load
    throws on (Data.h:214-221) with the reference's message; odd-but-legal tokens ("+3:1e-2", "7: 2", tabs) parse."""
    from libfm_amd import data as D
    import datagen
    ent, rp, y = datagen.ragged_real(500, 300, 14, seed=5, empty_every=17)
    p = str(tmp_path / "big.libfm")
    oracle.Data(ent, rp, y).write_libsvm(p)
    a, b = D.read_libsvm(p), D.read_libsvm_py(p)
    for x, z in zip(a, b):
        assert np.array_equal(x, z)
    assert np.array_equal(a[0]["id"], ent["id"]) and np.array_equal(a[1], rp)
    q = str(tmp_path / "odd.libfm")
    with open(q, "w") as f:
        f.write("1e0\t+3:1e-2 7: 2\t\n  -0.5 4294967295:1   # note\n")
    e2, r2, y2 = D.read_libsvm(q)
    assert list(y2) == [1.0, -0.5] and list(e2["id"]) == [3, 7, 4294967295] and list(r2) == [0, 2, 3]
    assert np.allclose(e2["value"], [0.01, 2.0, 1.0])
    for bad, needle in (("1 3:1 x\n", "cannot parse line"), ("abc\n", "cannot parse line"), ("1 3:1\r\n", "cannot parse line"),
                        ("1 2:\n", "cannot parse line")):
        with open(q, "w") as f:
            f.write(bad)
        with pytest.raises(ValueError, match=needle):
            D.read_libsvm(q)
    with pytest.raises(OSError, match="unable to open"):
        D.read_libsvm(str(tmp_path / "missing"))


@pytest.mark.skipif(not os.path.exists(CONVERT), reason="oracle/_ref/convert not built (needs /root/reference)")
def test_reads_what_the_reference_convert_tool_writes(tmp_path, oracle):
    from libfm_amd import data as D
    ent, rp, y = datagen.movielens_shaped(50, 40, 300, seed=1)
    txt = str(tmp_path / "d.libfm")
    oracle.Data(ent, rp, y).write_libsvm(txt)
    pre = str(tmp_path / "d")
    subprocess.run([CONVERT, "--ifile", txt, "--ofilex", pre + ".x", "--ofiley", pre + ".y"], check=True, capture_output=True)
    e2, r2, y2 = D.load(pre)
    assert np.array_equal(e2, ent) and np.array_equal(r2, rp) and np.array_equal(y2, y)
    # and our writer produces byte-identical files
    D.write_binary(str(tmp_path / "w"), ent, rp, y)
    assert open(pre + ".x", "rb").read() == open(str(tmp_path / "w") + ".x", "rb").read()
    assert open(pre + ".y", "rb").read() == open(str(tmp_path / "w") + ".y", "rb").read()


def test_binary_reader_of_the_c_abi(tmp_path, oracle):
    """fmx_read_binary (C-ABI, host side): .x + .y; only .xt + .y (rows rebuilt from the transpose); the legacy names;
    ragged rows with empty ones; and the files it must refuse."""
    from libfm_amd import data as D
    ent, rp, y = datagen.ragged_real(120, 300, 9, seed=4, empty_every=7)
    pre = str(tmp_path / "r")
    D.write_binary(pre, ent, rp, y, num_cols=120)
    e2, r2, y2, nf = D.read_binary(pre)
    assert np.array_equal(e2, ent) and np.array_equal(r2, rp) and np.array_equal(y2, y) and nf == 120
    assert np.array_equal(D.read_binary_x(pre + ".x")[0], e2)                       # the numpy reader agrees
    # transpose only: what an als / mcmc run keeps on disk (libfm.cpp:143-147)
    t_ent, t_ptr = D.transpose(ent, rp, 120)
    pt = str(tmp_path / "t")
    D.write_binary_matrix(pt + ".xt", t_ent, t_ptr, num_cols=len(rp) - 1)
    os.link(pre + ".y", pt + ".y")
    e3, r3, y3, nf3 = D.read_binary(pt)
    assert np.array_equal(r3, rp) and np.array_equal(y3, y) and nf3 == 120
    for r in range(len(rp) - 1):                                                    # same rows, entries in ascending id order
        a, b = int(rp[r]), int(rp[r + 1])
        order = np.argsort(ent["id"][a:b], kind="stable")
        assert np.array_equal(e3[a:b], ent[a:b][order])
    # legacy names (Data.h:120-121)
    pl = str(tmp_path / "legacy")
    os.link(pre + ".x", pl + ".data"); os.link(pre + ".y", pl + ".target")
    assert np.array_equal(D.read_binary(pl)[0], ent)
    # refusals
    with pytest.raises(ValueError, match="neither"):
        D.read_binary(str(tmp_path / "missing"))
    bad = str(tmp_path / "bad")
    raw = bytearray(open(pre + ".x", "rb").read())
    raw[0] = 9                                                                       # wrong file id (fmatrix.h:188)
    open(bad + ".x", "wb").write(raw); os.link(pre + ".y", bad + ".y")
    with pytest.raises(ValueError, match="not a libFM binary matrix"):
        D.read_binary(bad)
    trunc = str(tmp_path / "trunc")
    open(trunc + ".x", "wb").write(open(pre + ".x", "rb").read()[:-5]); os.link(pre + ".y", trunc + ".y")
    with pytest.raises(ValueError, match="truncated"):
        D.read_binary(trunc)
    short = str(tmp_path / "short")
    os.link(pre + ".x", short + ".x")
    open(short + ".y", "wb").write(np.array([1, 4, 5], dtype="<u4").tobytes() + np.zeros(5, dtype="<f4").tobytes())
    with pytest.raises(ValueError, match="row count differs"):
        D.read_binary(short)


TRANSPOSE = os.path.join(ROOT, "oracle", "_ref", "transpose")


@pytest.mark.skipif(not (os.path.exists(CONVERT) and os.path.exists(TRANSPOSE)), reason="oracle/_ref/convert / transpose not built (needs /root/reference)")
def test_binary_reader_on_the_reference_tools_output(tmp_path, oracle):
    """convert + transpose of the REFERENCE write the files; fmx_read_binary reads .x and, alone, .xt"""
    from libfm_amd import data as D
    ent, rp, y = datagen.movielens_shaped(60, 45, 400, seed=6)
    txt = str(tmp_path / "d.libfm")
    oracle.Data(ent, rp, y).write_libsvm(txt)
    pre = str(tmp_path / "d")
    subprocess.run([CONVERT, "--ifile", txt, "--ofilex", pre + ".x", "--ofiley", pre + ".y"], check=True, capture_output=True)
    subprocess.run([TRANSPOSE, "--ifile", pre + ".x", "--ofile", pre + ".xt"], check=True, capture_output=True)
    e2, r2, y2, nf = D.read_binary(pre)
    assert np.array_equal(e2, ent) and np.array_equal(r2, rp) and np.array_equal(y2, y) and nf == int(ent["id"].max()) + 1
    only_t = str(tmp_path / "t")
    os.link(pre + ".xt", only_t + ".xt"); os.link(pre + ".y", only_t + ".y")
    e3, r3, y3, nf3 = D.read_binary(only_t)
    assert np.array_equal(e3, ent) and np.array_equal(r3, rp) and nf3 == nf       # these rows already have ascending ids


LIBFM = os.path.join(ROOT, "oracle", "_ref", "libFM")


@pytest.mark.skipif(not os.path.exists(LIBFM), reason="oracle/_ref/libFM not built (needs /root/reference)")
def test_model_file_round_trip_is_byte_identical_to_the_reference_writer(tmp_path, oracle):
    """-save_model text format (fm_model.h:132-154): read the stock binary's file, write it again, same bytes."""
    from libfm_amd import learner as L
    ent, rp, y = datagen.movielens_shaped(30, 20, 200, seed=2)
    txt = str(tmp_path / "d.libfm")
    oracle.Data(ent, rp, y).write_libsvm(txt)
    mf = str(tmp_path / "model")
    subprocess.run([LIBFM, "-task", "r", "-train", txt, "-test", txt, "-dim", "1,1,4", "-iter", "3", "-method", "sgd",
                    "-learn_rate", "0.01", "-init_stdev", "0.1", "-seed", "7", "-save_model", mf],
                   check=True, capture_output=True)
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor = 50, 4
    assert fm.load_model(mf)
    out = str(tmp_path / "model2")
    fm.save_model(out)
    assert open(mf).read() == open(out).read()
    fm2 = L.FMModel()
    fm2.num_attribute, fm2.num_factor = 50, 5
    assert not fm2.load_model(mf)                      # wrong k: "malformed model file" (libfm.cpp:264-267)
