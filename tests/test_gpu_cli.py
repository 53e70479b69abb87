"""GPU: the libFM-flag command line (libfm_amd/cli.py) end to end on BASELINE.json configs[0]: same flags, same
seed, same data as the STOCK reference binary run that produced tests/golden/c1_ml100k_shaped.npz -- the #Iter
lines, the -out file and the -save_model file must agree with what the reference printed (6 significant digits)."""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def test_cli_matches_stock_libfm_on_config1(tmp_path, oracle):
    from libfm_amd import cli
    z = np.load(os.path.join(GOLDEN_DIR, "c1_ml100k_shaped.npz"))
    trf, tef = str(tmp_path / "ml.train.libfm"), str(tmp_path / "ml.test.libfm")
    oracle.Data(z["train_entries"], z["train_row_ptr"].astype(np.uint64), z["train_target"]).write_libsvm(trf)
    oracle.Data(z["test_entries"], z["test_row_ptr"].astype(np.uint64), z["test_target"]).write_libsvm(tef)
    out, model = str(tmp_path / "pred"), str(tmp_path / "model")
    # the stock command line (z["cmdline"]) with our file names, plus the GPU mode that follows the reference order
    argv = ["-task", "r", "-train", trf, "-test", tef, "-dim", "1,1,8", "-iter", "20", "-method", "sgd",
            "-learn_rate", "0.01", "-regular", "0,0,0.01", "-init_stdev", "0.1", "-seed", "42",
            "-out", out, "-save_model", model, "-gpu_mode", "sequential"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(argv) == 0
    lines = [[float(x.split("=")[1]) for x in ln.split("\t")[1:3]] for ln in buf.getvalue().splitlines() if ln.startswith("#Iter=")]
    np.testing.assert_allclose(np.array(lines), z["stdout_iters"], rtol=1e-4)
    np.testing.assert_allclose(np.loadtxt(out), z["out_pred"], rtol=1e-4, atol=1e-5)
    txt = open(model).read().splitlines()
    assert txt[0] == "#global bias W0" and txt[2] == "#unary interactions Wj"
    n = 943 + 1682
    np.testing.assert_allclose(float(txt[1]), float(z["model_w0"]), rtol=1e-4)
    v = np.array([[float(x) for x in ln.split()] for ln in txt[4 + n:4 + 2 * n]]).T
    np.testing.assert_allclose(v, z["model_v"], rtol=1e-4, atol=2e-5)


def test_cli_error_convention(capsys):
    from libfm_amd import cli
    assert cli.main(["-task", "r", "-train", "/nonexistent", "-test", "/nonexistent", "-method", "sgd", "-learn_rate", "0.1"]) == 0
    assert "ERROR:" in capsys.readouterr().err            # "ERROR: ..." on stderr, exit status 0 (libfm.cpp:436-441)
    assert cli.main(["-task", "r", "-bogus", "1"]) == 0
    assert "does not exist" in capsys.readouterr().err


def test_cli_als_with_meta_groups(tmp_path, oracle):
    """`-method als -meta <file> -regular 'r0,w_1,w_2,v_1,v_2'` (libfm.cpp:199-242, 353-363): same seed, same data as the
    reference run behind tests/golden/als_reg_ml_groups.npz -> the same -out file and model."""
    from libfm_amd import cli
    z = np.load(os.path.join(GOLDEN_DIR, "als_reg_ml_groups.npz"))
    trf, tef, meta = str(tmp_path / "tr.libfm"), str(tmp_path / "te.libfm"), str(tmp_path / "meta")
    oracle.Data(z["train_entries"], z["train_row_ptr"], z["train_target"]).write_libsvm(trf)
    oracle.Data(z["test_entries"], z["test_row_ptr"], z["test_target"]).write_libsvm(tef)
    with open(meta, "w") as f:
        f.write("".join("%d\n" % g for g in z["group"]))
    reg = [float(z["reg"][0])] + [float(x) for x in z["w_lambda_g"]] + [float(x) for x in z["v_lambda_g"]]
    out, model = str(tmp_path / "pred"), str(tmp_path / "model")
    argv = ["-task", "r", "-train", trf, "-test", tef, "-dim", "%d,%d,%d" % (int(z["k0"]), int(z["k1"]), int(z["k"])),
            "-iter", str(int(z["iters"])), "-method", "als", "-meta", meta, "-regular", ",".join(repr(x) for x in reg),
            "-init_stdev", repr(float(z["init_stdev"])), "-seed", str(int(z["seed"])), "-out", out, "-save_model", model]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(argv) == 0
    assert "#groups=2" in buf.getvalue()
    np.testing.assert_allclose(np.loadtxt(out), z["pred_out"], rtol=1e-4, atol=5e-5)
    txt = open(model).read().splitlines()
    n = int(z["n"])
    np.testing.assert_allclose(float(txt[1]), float(z["final_w0"]), rtol=1e-4, atol=1e-5)
    w = np.array([float(x) for x in txt[3:3 + n]])
    np.testing.assert_allclose(w, z["final_w"], rtol=1e-4, atol=2e-5)
