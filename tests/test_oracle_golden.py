"""Pin the CPU oracle (oracle/neunet_oracle.py) against golden vectors produced by the REAL
reference (tools/gen_golden.py, run in the build container).  CPU-only."""
import numpy as np
import pytest

from oracle import neunet_oracle as O

TOL = dict(rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["linear_2d", "linear_3d", "linear_nobias"])
def test_linear(golden, name):
    g = golden(name)
    b = g.get("b")
    np.testing.assert_allclose(O.linear_forward(g["X"], g["W"], b), g["O"], **TOL)
    dX, dW, db = O.linear_backward(g["X"], g["W"], b, g["dO"])
    np.testing.assert_allclose(dX, g["dX"], **TOL)
    np.testing.assert_allclose(dW, g["dW"], rtol=1e-5, atol=1e-5)
    assert dW.shape == g["W"].shape
    if b is not None:
        assert db.shape == (1, 40)
        np.testing.assert_allclose(db, g["db"], rtol=1e-5, atol=1e-5)


def test_relu(golden):
    g = golden("relu")
    y = O.relu_forward(g["X"])
    np.testing.assert_array_equal(y, g["Y"])
    np.testing.assert_array_equal(O.relu_backward(y, g["dY"]), g["dX"])


@pytest.mark.parametrize("name", ["swish_b1.0", "swish_b1.5"])
def test_swish(golden, name):
    g = golden(name)
    beta = float(g["beta"])
    np.testing.assert_allclose(O.swish_forward(g["X"], beta), g["Y"], **TOL)
    np.testing.assert_allclose(O.swish_backward(g["X"], g["dY"], beta), g["dX"], **TOL)


@pytest.mark.parametrize("name", ["swiglu_2d", "swiglu_3d"])
def test_swiglu(golden, name):
    g = golden(name)
    beta = float(g["beta"])
    np.testing.assert_allclose(O.swiglu_forward(g["X"], beta), g["Y"], **TOL)
    np.testing.assert_allclose(O.swiglu_backward(g["X"], g["dY"], beta), g["dX"], **TOL)


@pytest.mark.parametrize("name", ["softmax_last", "softmax_axis1_4d", "softmax_axis1_2d"])
def test_softmax(golden, name):
    g = golden(name)
    ax = int(g["axis"])
    y = O.softmax_forward(g["X"], ax)
    np.testing.assert_allclose(y, g["Y"], **TOL)
    np.testing.assert_allclose(O.softmax_backward(y, g["dY"], ax), g["dX"], **TOL)


@pytest.mark.parametrize("name", ["ce_mean", "ce_sum", "ce_none", "ce_mean_ign", "ce_sum_ign",
                                  "ce_none_ign", "ce_mean_pad0", "ce_mean_small"])
def test_cross_entropy(golden, name):
    g = golden(name)
    loss, dl = O.cross_entropy_forward_backward(g["logits"], g["labels"], None, int(g["ignore_index"]),
                                                str(g["reduction"]))
    np.testing.assert_allclose(np.reshape(loss, g["loss"].shape), g["loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dl, g["dlogits"], rtol=1e-5, atol=1e-7)
    ign = g["labels"] == int(g["ignore_index"])
    assert np.all(dl[ign] == 0)


@pytest.mark.parametrize("name", ["rmsnorm_2d", "rmsnorm_3d_bias"])
def test_rmsnorm(golden, name):
    g = golden(name)
    b = g.get("b")
    Y, _, _ = O.rmsnorm_forward(g["X"], g["w"], b, float(g["eps"]))
    np.testing.assert_allclose(Y, g["Y"], **TOL)
    dX, dw, db = O.rmsnorm_backward(g["X"], g["w"], b is not None, g["dY"], float(g["eps"]))
    np.testing.assert_allclose(dX, g["dX"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dw, g["dw"], rtol=1e-5, atol=1e-5)
    if b is not None:
        np.testing.assert_allclose(db, g["db"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["conv2d_s2p1d2", "conv2d_s2_uncovered", "conv2d_pad4", "conv2d_c5_l1",
                                  "conv2d_c5_l2"])
def test_conv2d(golden, name):
    g = golden(name)
    st, pad, dil = tuple(g["stride"]), tuple(int(p) for p in g["padding"]), tuple(g["dilation"])
    geo = O.conv2d_geometry(g["X"].shape[2:], g["W"].shape[2:], st, pad, dil)
    assert tuple(geo["pad"]) == tuple(g["padding4"])
    out = O.conv2d_forward(g["X"], g["W"], g["b"], st, pad, dil)
    assert out.shape == g["O"].shape
    np.testing.assert_allclose(out, g["O"], rtol=1e-5, atol=1e-5)
    dX, dW, db = O.conv2d_backward(g["X"], g["W"], True, g["dO"], st, pad, dil)
    np.testing.assert_allclose(dX, g["dX"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dW, g["dW"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db, g["db"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["adam_wd0", "adam_wd1e-2", "adamw_wd0", "adamw_wd1e-2"])
def test_adam(golden, name):
    g = golden(name)
    fn = O.adamw_step if name.startswith("adamw") else O.adam_step
    n = int(g["n_tensors"])
    p = [g[f"p0_{i}"].copy() for i in range(n)]
    m = [np.zeros_like(a) for a in p]
    v = [np.zeros_like(a) for a in p]
    for s in range(3):
        for i in range(n):
            m[i], v[i] = fn(p[i], g[f"g{s}_{i}"], m[i], v[i], s + 1, float(g["lr"]), (0.9, 0.999), 1e-8,
                            float(g["wd"]))
            np.testing.assert_allclose(p[i], g[f"p{s + 1}_{i}"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(m[i], g[f"m{s + 1}_{i}"], rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(v[i], g[f"v{s + 1}_{i}"], rtol=1e-6, atol=1e-9)


def test_linear_swish(golden):
    g = golden("linear_swish")
    beta = float(g["beta"])
    y, _ = O.linear_swish_forward(g["X"], g["W"], g["b"], beta)
    np.testing.assert_allclose(y, g["Y"], **TOL)
    dX, dW, db = O.linear_swish_backward(g["X"], g["W"], g["b"], g["dY"], beta)
    np.testing.assert_allclose(dX, g["dX"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dW, g["dW"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db, g["db"], rtol=1e-5, atol=1e-5)


def test_mlp_c1_trajectory(golden):
    """README quick-start loop (C1): losses, argmax (bit-exact), grads and weights after 3 Adam steps."""
    g = golden("mlp_c1")
    st = O.MLPState(g["W1"], g["b1"], g["W2"], g["b2"], lr=1e-3)
    for s in range(3):
        loss, logits, grads = st.step(g["X"][s], g["Y"][s])
        assert abs(float(loss) - g["losses"][s]) < 1e-5
        np.testing.assert_array_equal(np.argmax(logits, axis=1).astype(np.int32), g["argmax"][s])
        if s == 0:
            np.testing.assert_allclose(grads[0][::8], g["dW1_step0_rows"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(grads[1], g["db1_step0"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(grads[2], g["dW2_step0"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(grads[3], g["db2_step0"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st.p[0][::8], g["W1_final_rows"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.p[1], g["b1_final"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.p[2], g["W2_final"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.p[3], g["b2_final"], rtol=1e-5, atol=1e-6)


# ---- embedding (last-write-wins grad), multi-head attention, GPT-tiny step -----------------
def test_embedding(golden):
    g = golden("embedding")
    np.testing.assert_array_equal(O.embedding_forward(g["W"], g["ids"]), g["out"])
    np.testing.assert_array_equal(O.embedding_backward(g["W"].shape, g["ids"], g["grad"]), g["dW"])


def test_mha(golden):
    g = golden("mha")
    m = O.MHA(g["Wq"], g["bq"], g["Wk"], g["bk"], g["Wv"], g["bv"], g["Wo"], g["bo"], int(g["n_heads"]))
    y, attn = m.forward(g["X"], g["mask"])
    np.testing.assert_allclose(attn, g["attn"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y, g["Y"], rtol=1e-5, atol=1e-5)
    dx, grads = m.backward(g["dY"])
    np.testing.assert_allclose(dx, g["dX"], rtol=1e-4, atol=1e-5)
    for name, dW, db in zip("qkvo", grads[0::2], grads[1::2]):
        np.testing.assert_allclose(dW, g[f"dW{name}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(db, g[f"db{name}"], rtol=1e-4, atol=1e-5)
    # the structured mask (key padding + causal) used by the HIP path equals the notebook's dense mask
    kv = g["key_valid"]
    T = kv.shape[1]
    dense = (kv[:, None, :] & np.tril(np.ones((T, T), dtype=np.int32))[None]).astype(np.int32)
    np.testing.assert_array_equal(dense, g["mask"])


def gpt_from_golden(g):
    """Map the notebook's Module.parameters() order onto the oracle's GPTTiny (22 params per layer)."""
    V, D, H, F, L = [int(v) for v in g["cfg"]]
    P = lambda i: g[f"p{i}"]  # noqa: E731
    layers, idx = [], 1
    for _ in range(L):
        attn = [P(idx + j) for j in range(8)]
        ffn = [P(idx + 16 + j) for j in range(4)]
        layers.append({"attn": attn, "ffn": ffn, "norm1": P(idx + 20), "norm2": P(idx + 21), "base": idx})
        idx += 22
    return O.GPTTiny(P(0), layers, P(idx), P(idx + 1), H, pad_idx=0, max_len=64), idx


def test_gpt_tiny_step(golden):
    g = golden("gpt_tiny")
    model, out_idx = gpt_from_golden(g)
    batch = g["batch"]
    loss, logits, grads = model.forward_backward(batch[:, :-1], batch[:, 1:])
    np.testing.assert_allclose(logits, g["logits"], rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    np.testing.assert_allclose(grads["emb"], g["g0"], rtol=1e-4, atol=1e-6)
    for L, gl in zip(model.layers, grads["layers"]):
        b = L["base"]
        for j in range(8):
            np.testing.assert_allclose(gl["attn"][j], g[f"g{b + j}"], rtol=1e-3, atol=1e-6)
            assert not bool(g[f"has_grad{b + 8 + j}"])       # cross_attn is never called
        for j in range(4):
            np.testing.assert_allclose(gl["ffn"][j], g[f"g{b + 16 + j}"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(gl["norm1"], g[f"g{b + 20}"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(gl["norm2"], g[f"g{b + 21}"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(grads["Wout"], g[f"g{out_idx}"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(grads["bout"], g[f"g{out_idx + 1}"], rtol=1e-4, atol=1e-6)


# ---- the remaining conv-classifier ops and the full config-5 step -----------------------------------
def test_maxpool_dilated(golden):
    """MaxPool2d with dilation > 1 (maxpool2d.py:170-220): the oracle's tap formulation == the reference's NaN-kernel one."""
    g = golden("maxpool_dilated")
    X = g["X"]
    for tag in ("k2s1p0d2", "k3s2p2d2", "k2s2p1d3"):
        ks, st, pad, dil = [int(v) for v in g[f"{tag}_cfg"]]
        y, arg = O.maxpool2d_forward(X, (ks, ks), (st, st), (pad, pad), (dil, dil))
        np.testing.assert_array_equal(y, g[f"{tag}_Y"])
        dX = O.maxpool2d_backward(X.shape, arg, g[f"{tag}_dY"], (ks, ks), (st, st), (pad, pad), (dil, dil))
        np.testing.assert_allclose(dX, g[f"{tag}_dX"], rtol=1e-6, atol=1e-6)


def test_vision_ops(golden):
    g = golden("vision_ops")
    X = g["X"]
    f = O.leaky_relu_forward(X)
    np.testing.assert_allclose(f, g["leaky_Y"], **TOL)
    np.testing.assert_allclose(O.leaky_relu_backward(f, g["leaky_dY"]), g["leaky_dX"], **TOL)
    f = O.sigmoid_forward(X)
    np.testing.assert_allclose(f, g["sigmoid_Y"], **TOL)
    np.testing.assert_allclose(O.sigmoid_backward(f, g["sigmoid_dY"]), g["sigmoid_dX"], **TOL)
    for tag in ("pool22", "pool32p1", "pool21_overlap"):
        ks, st, pad = [int(v) for v in g[f"{tag}_cfg"]]
        y, arg = O.maxpool2d_forward(X, (ks, ks), (st, st), (pad, pad))
        np.testing.assert_array_equal(y, g[f"{tag}_Y"])
        np.testing.assert_allclose(O.maxpool2d_backward(X.shape, arg, g[f"{tag}_dY"], (ks, ks), (st, st), (pad, pad)),
                                   g[f"{tag}_dX"], rtol=1e-6, atol=1e-6)
    rm0, rv0 = np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32)
    y, cache, rm, rv = O.batchnorm2d_forward(X, g["bn_w"], g["bn_b"], rm0, rv0)
    np.testing.assert_allclose(y, g["bn_Y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rm, g["bn_rm"], **TOL)
    np.testing.assert_allclose(rv, g["bn_rv"], **TOL)
    dX, dw, db = O.batchnorm2d_backward(X, g["bn_w"], cache, g["bn_dY"])
    np.testing.assert_allclose(dX, g["bn_dX"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dw, g["bn_dw"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db, g["bn_db"], rtol=1e-5, atol=1e-5)
    ye, _, _, _ = O.batchnorm2d_forward(X, g["bn_w"], g["bn_b"], rm, rv, training=False)
    np.testing.assert_allclose(ye, g["bn_Yeval"], rtol=1e-5, atol=1e-5)
    y, cache, _, _ = O.batchnorm2d_forward(X, None, None, rm0, rv0)
    np.testing.assert_allclose(y, g["bn_noaffine_Y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(O.batchnorm2d_backward(X, None, cache, g["bn_noaffine_dY"])[0], g["bn_noaffine_dX"],
                               rtol=1e-4, atol=1e-5)
    loss, dP = O.mse_forward_backward(g["mse_P"], g["mse_T"])
    assert abs(float(loss) - float(g["mse_loss"])) < 1e-6
    np.testing.assert_allclose(dP, g["mse_dP"], **TOL)


def test_conv_classifier_step(golden):
    g = golden("conv_classifier")
    n = int(g["n_params"])
    model = O.ConvClassifier([g[f"p{i}"] for i in range(n)])
    m = [np.zeros_like(a) for a in model.p]
    v = [np.zeros_like(a) for a in model.p]
    for s in range(2):
        loss, out, grads = model.forward_backward(g["X"][s], g["T"][s])
        # step 2 runs on parameters that went through one Adam step: Adam's first update is lr*sign(g), so
        # gradients at rounding-noise level (e.g. parts of conv biases) move by a full +-lr in either
        # implementation and the second forward agrees only to ~lr
        assert abs(float(loss) - g["losses"][s]) < (1e-6 if s == 0 else 2e-4)
        np.testing.assert_allclose(out, g["outs"][s], rtol=1e-4, atol=1e-5 if s == 0 else 1e-3)
        if s == 0:
            for i in range(n):
                np.testing.assert_allclose(grads[i], g[f"g{i}"], rtol=1e-3, atol=1e-6, err_msg=f"grad {i}")
        for i in range(n):
            m[i], v[i] = O.adam_step(model.p[i], grads[i].astype(np.float32), m[i], v[i], s + 1, 1e-3)
    # running statistics after step 2 inherit the same +-lr noise through the conv biases (a bias shift moves the
    # channel mean one-for-one); step-1 statistics are pinned exactly by test_vision_ops
    np.testing.assert_allclose(model.rm, g["rm"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(model.rv, g["rv"], rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_cross_entropy_class_weights(golden, reduction):
    """CrossEntropyLoss(weight=w) of the reference (losses.py:93-118), int64 labels, ignored rows."""
    g = golden("ce_weighted")
    loss, dl = O.cross_entropy_forward_backward(g["logits"], g["labels"], g["weight"], int(g["ignore_index"]), reduction)
    np.testing.assert_allclose(np.reshape(loss, -1), g[f"loss_{reduction}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dl, g[f"dlogits_{reduction}"], rtol=1e-5, atol=1e-7)
    assert np.all(dl[g["labels"] == int(g["ignore_index"])] == 0)


def test_reference_written_checkpoint_matches_the_gpt_fixture(golden):
    """tests/golden/gpt_tiny_state.pkl was written by the REFERENCE (neunet.save(model.state_dict()),
    neunet/__init__.py:26-29, nn/modules.py:76-86).  Its values, walked in key order, are the p{i} arrays of the gpt_tiny
    fixture in Module.parameters() order -- so the GPU test that loads it through load_state_dict starts from the
    reference's weights."""
    import os
    import pickle
    from conftest import GOLDEN
    g = golden("gpt_tiny")
    with open(os.path.join(GOLDEN, "gpt_tiny_state.pkl"), "rb") as f:
        sd = pickle.load(f)
    assert len(sd) == int(g["n_params"])
    keys = list(sd)
    assert keys[0] == "decoder.token_embedding.weight" and keys[-1] == "decoder.fc_out.bias"
    assert "decoder.layers.0.self_attn.wq.weight" in sd and "decoder.layers.1.cross_attn.fc.bias" in sd
    for i, k in enumerate(keys):
        np.testing.assert_array_equal(sd[k], g[f"p{i}"], err_msg=k)


def test_fixtures_regenerate_bit_for_bit(tmp_path):
    """Every generator in tools/gen_golden.py seeds the global np.random that the reference layers initialise from, so
    re-running it against /root/reference reproduces every committed fixture exactly (build container only)."""
    import os
    import pickle
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    if not os.path.isdir("/root/reference/neunet"):
        pytest.skip("/root/reference is only present in the build container")
    out = str(tmp_path / "golden")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden.py"), "--out", out], check=True,
                   stdout=subprocess.DEVNULL, timeout=600)
    names = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert names == sorted(f for f in os.listdir(out) if f.endswith(".npz"))
    for f in names:
        a, b = np.load(os.path.join(GOLDEN, f), allow_pickle=False), np.load(os.path.join(out, f), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{f}:{k}")
    sa = pickle.load(open(os.path.join(GOLDEN, "gpt_tiny_state.pkl"), "rb"))
    sb = pickle.load(open(os.path.join(out, "gpt_tiny_state.pkl"), "rb"))
    assert list(sa) == list(sb)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
