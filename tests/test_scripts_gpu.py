"""End-to-end through the command-line entry points on the GPU: train (custom data source) ->
checkpoint -> sample -> attention dump, the way a user of the reference's scripts/ would run them."""
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

BASE = ["CCO", "c1ccccc1", "CC(=O)O", "CCN(CC)CC", "C1CCCCC1", "CC(C)Cc1ccc(cc1)C(C)C(=O)O", "O=C(O)c1ccccc1O",
        "CCOC(=O)C", "c1ccncc1", "CC(C)O", "ClCCl", "FC(F)F", "CSC", "NC(=O)C", "C#N", "Brc1ccccc1"]


def test_train_sample_attention_scripts(tmp_path, monkeypatch):
    from scripts import attention, parsers, sample, train
    monkeypatch.chdir(tmp_path)
    mols = [BASE[i % len(BASE)] + "C" * (i // len(BASE)) for i in range(48)]
    pd.DataFrame(mols, columns=["mol"]).to_csv("train.txt", index=False)
    pd.DataFrame(mols[:24], columns=["mol"]).to_csv("test.txt", index=False)
    vocab = os.path.join(ROOT, "transvae_amd", "data", "vocab_zinc.json")
    args = parsers.train_parser().parse_args(
        ["--model", "transvae", "--data_source", "custom", "--train_mols_path", "train.txt", "--test_mols_path",
         "test.txt", "--vocab_path", vocab, "--batch_size", "12", "--batch_chunks", "2", "--epochs", "2",
         "--save_freq", "1", "--save_name", "tiny", "--d_model", "128", "--d_feedforward", "128"])
    train.train(args)
    assert os.path.exists("checkpoints/001_tiny.ckpt") and os.path.exists("checkpoints/002_tiny.ckpt")
    log = pd.read_csv("trials/log_tiny.txt")
    assert set(log["data_type"]) == {"train", "test"} and len(log) == 2 * (4 + 2)
    assert np.isfinite(log["tot_loss"]).all()

    # resume for one more epoch: weights, the epoch counter and the optimizer's step count (hence the Noam rate) carry over;
    # Adam's moments restart from zero, as in the reference (opt.py:38-39 only replaces the dict attribute)
    args2 = parsers.train_parser().parse_args(
        ["--model", "transvae", "--data_source", "custom", "--train_mols_path", "train.txt", "--test_mols_path",
         "test.txt", "--vocab_path", vocab, "--batch_size", "12", "--batch_chunks", "2", "--epochs", "1",
         "--save_freq", "1", "--save_name", "tiny", "--checkpoint", "checkpoints/002_tiny.ckpt"])
    train.train(args2)
    assert os.path.exists("checkpoints/003_tiny.ckpt")

    sargs = parsers.sample_parser().parse_args(
        ["--model", "transvae", "--model_ckpt", "checkpoints/003_tiny.ckpt", "--sample_mode", "rand",
         "--n_samples", "10", "--n_samples_per_batch", "5", "--save_path", "gen.csv"])
    sample.sample(sargs)
    assert len(pd.read_csv("gen.csv", keep_default_na=False)) == 10

    aargs = parsers.attn_parser().parse_args(
        ["--model", "transvae", "--model_ckpt", "checkpoints/003_tiny.ckpt", "--mols", "train.txt",
         "--n_samples", "12", "--batch_size", "6", "--batch_chunks", "2", "--save_path", "attn"])
    attention.calc_attention(aargs)
    sa, xa = np.load("attn_self_attn.npy"), np.load("attn_src_attn.npy")
    assert sa.shape == (12, 4, 4, 127, 127) and xa.shape == (12, 3, 4, 126, 127)
    np.testing.assert_allclose(sa.sum(-1), 1.0, atol=1e-4)          # every dumped row is a softmax
    np.testing.assert_allclose(xa.sum(-1), 1.0, atol=1e-4)


@pytest.mark.parametrize("model", ["rnnattn", "rnn"])
def test_rnn_train_script_and_shell_train(tmp_path, monkeypatch, model):
    """scripts/train.py --model rnn|rnnattn and VAEShell.train on the recurrent shells: the epoch end (optimizer state
    mirrored into the checkpoint through AdamOpt.sync_state_dict) runs, the checkpoint reloads through load_fn"""
    from scripts import parsers, train
    from transvae_amd.rnn_models import RNN, RNNAttn
    monkeypatch.chdir(tmp_path)
    mols = [BASE[i % len(BASE)] + "C" * (i // len(BASE)) for i in range(48)]
    pd.DataFrame(mols, columns=["mol"]).to_csv("train.txt", index=False)
    pd.DataFrame(mols[:24], columns=["mol"]).to_csv("test.txt", index=False)
    vocab = os.path.join(ROOT, "transvae_amd", "data", "vocab_zinc.json")
    args = parsers.train_parser().parse_args(
        ["--model", model, "--data_source", "custom", "--train_mols_path", "train.txt", "--test_mols_path",
         "test.txt", "--vocab_path", vocab, "--batch_size", "12", "--batch_chunks", "2", "--epochs", "2",
         "--save_freq", "1", "--save_name", "tiny", "--d_model", "128"])
    train.train(args)
    assert os.path.exists("checkpoints/002_tiny.ckpt")
    cls = RNNAttn if model == "rnnattn" else RNN
    vae = cls(load_fn="checkpoints/002_tiny.ckpt")
    assert vae.n_epochs == 2 and "state" in vae.current_state["optimizer_state_dict"]
    # the shell's own train() on the reloaded model, one more epoch, saved and reloaded again
    vae.train(np.array(mols)[:, None], np.array(mols[:24])[:, None], epochs=1, save=True, log=False)
    assert os.path.exists("checkpoints/003_tiny.ckpt")
    again = cls(load_fn="checkpoints/003_tiny.ckpt")
    for (k, a), (_, b) in zip(vae.model.state_dict().items(), again.model.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    assert again.greedy_decode(torch.zeros(3, again.params["d_latent"])).shape == (3, 125)


@pytest.mark.parametrize("mode", ["rand", "k_high_entropy"])
def test_sample_script_slab_decode_equals_per_batch_loop(tmp_path, monkeypatch, mode):
    """scripts/sample.py draws the latents batch by batch (the reference's order of RNG calls, scripts/sample.py:47-51)
    and decodes them together in slabs: for the same seed the CSV equals the literal per-batch loop's"""
    from scripts import parsers, sample
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    monkeypatch.chdir(tmp_path)
    cd, od, w = synth.load_vocab("zinc")
    vae = TransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": 12, "BATCH_CHUNKS": 2},
                   name="slab", **synth.ARCHS["trans1x-128"])
    vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 0))
    vae.current_state["model_state_dict"] = vae.model.state_dict()
    vae.save(vae.current_state, "slab.ckpt", use_name=False)
    mols = [BASE[i % len(BASE)] + "C" * (i // len(BASE)) for i in range(500)]     # calc_mems works on full batches of 500
    pd.DataFrame(mols, columns=["mol"]).to_csv("mols.txt", index=False)
    outs = []
    for per_batch in ("1", "0"):
        monkeypatch.setenv("TVAE_SAMPLE_PER_BATCH", per_batch)
        monkeypatch.setattr(TransVAE, "DECODE_SLAB", 64)             # several slabs, the last one partial
        torch.manual_seed(5)
        np.random.seed(5)
        argv = ["--model", "transvae", "--model_ckpt", "slab.ckpt", "--sample_mode", mode, "--n_samples", "230",
                "--n_samples_per_batch", "25", "--save_path", "gen%s.csv" % per_batch]
        if mode != "rand":
            argv += ["--mols", "mols.txt", "--entropy_cutoff=-1e9", "--k", "7"]
        sample.sample(parsers.sample_parser().parse_args(argv))
        outs.append(pd.read_csv("gen%s.csv" % per_batch, keep_default_na=False)["mol"].tolist())
    assert len(outs[0]) == 250 and outs[0] == outs[1]


def test_repeated_train_calls_continue_the_dropout_stream(tmp_path, monkeypatch):
    """train(epochs=1) twice must not replay the first call's masks: the second call continues {seed, step, pass}
    (the reference continues torch's global generator); a process that never seeded gets seeded exactly once"""
    from transvae_amd import ops, synth
    from transvae_amd.trans_models import TransVAE
    monkeypatch.chdir(tmp_path)
    cd, od, w = synth.load_vocab("zinc")
    vae = TransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": 12, "BATCH_CHUNKS": 2},
                   name="rep", **synth.ARCHS["trans1x-128"])
    mols = np.array([BASE[i % len(BASE)] + "C" * (i // len(BASE)) for i in range(24)])[:, None]
    dev = torch.device("cuda:0")
    ops._seeded.discard((dev.type, dev.index))                    # as in a fresh process
    vae.train(mols, mols[:12], epochs=1, save=False, log=False)
    s1 = ops.rng_snapshot(dev)
    assert ops.rng_seeded(dev) and s1["step"] == 2 and s1["pass"] >= 2
    vae.train(mols, mols[:12], epochs=1, save=False, log=False)
    s2 = ops.rng_snapshot(dev)
    assert s2["seed"] == s1["seed"] and s2["step"] == 4 and s2["pass"] > s1["pass"]


CONTRACT_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _run_bench(*flags, env=None):
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def test_bench_prints_the_contract_line():
    """bench.py as the driver runs it (shortened): one JSON line with the contract's keys, the metric and config of
    BASELINE.json, a roofline object for the dominant kernel and the CPU baseline."""
    import json
    out = _run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--decode-n", "512", "--no-extra-modes")
    for k in CONTRACT_KEYS:
        assert k in out, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert out["metric"] == base["metric"] and out["unit"] == "tokens/s" and out["n_gpus"] == 1
    assert out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic" and "workload" in out["config"]
    assert abs(out["value"] - 500 * 126 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0.2 < rf["frac"] < 1.0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "tokens/s"
    assert out["value"] > 50 * cb["value"]              # a GPU step that is not faster than the host has fallen off the HIP path
    assert out["decode_mols_per_s"] > 0


def test_bench_roofline_only():
    out = _run_bench("--roofline-only", "--dtype", "f32x3")
    assert list(out) == ["roofline"] and "gemm3_dma_kernel" in out["roofline"]["kernel"]


def test_bench_data_parallel_path_over_rccl_with_one_rank():
    """The N > 1 code path of bench.py (RCCL process group, flat gradient arena, bucketed async all-reduce overlapped
    with backward, max-over-ranks timing, final gather of token ids) on the real `nccl` backend -- with the single
    rank a one-GPU box allows.  Two ranks over gloo are tests/test_parallel_gpu.py; N = 2, 4, 8 over xGMI are the driver's."""
    out = _run_bench("--steps", "2", "--warmup", "1", "--decode-n", "256", "--no-extra-modes", "--no-cpu-baseline",
                     env={"TVAE_BENCH_FORCE_DP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["decode_mols_per_s"] > 0
    assert out["final_loss"] == out["final_loss"] and 0 < out["final_loss"] < 100        # finite, sane


def test_bench_gpus_2_launches_its_own_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command, with N = 2): the
    script starts two ranks itself and reports n_gpus == 2.  On a one-GPU box the two ranks share cuda:0 over gloo
    (TVAE_BENCH_REHEARSE=1; numbers meaningless) -- trans4x-256, 100 molecules per rank, the data-parallel step, the
    sharded decode and the gather of token ids all run."""
    out = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "100", "--chunks", "1", "--decode-n", "128",
                     "--no-extra-modes", "--no-cpu-baseline", env={"TVAE_BENCH_REHEARSE": "1"})
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 200 and out["config"]["parallelism"] == "dp2"
    dp = out["data_parallel"]
    assert len(dp["ms_per_step_per_rank"]) == 2 and len(dp["allreduce_wait_ms_per_step_per_rank"]) == 2
    assert dp["gradient_bytes_per_step"] >= 4 * 7876205 and dp["buckets"] >= 2
    assert abs(out["value"] - 2 * 100 * 126 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert 0 < out["final_loss"] < 100 and out["decode_mols_per_s"] > 0


def test_bench_refuses_a_world_size_it_cannot_run():
    "without the rehearsal switch --gpus 2 on a one-GPU box is an error, never a line that says n_gpus 1 or 2"
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=120, env={k: v for k, v in os.environ.items() if k not in ("TVAE_BENCH_REHEARSE", "WORLD_SIZE")})
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr
    # a launcher that disagrees with the flag is refused too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=120, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "disagree" in r.stderr


def test_config5_entropy_sampling_and_attention_dump_on_trans4x256_pubchem_pp(tmp_path, monkeypatch):
    """BASELINE config 5 at its own size, end to end through the scripts: trans4x-256, PubChem vocabulary, property head;
    scripts/sample.py in high_entropy and k_high_entropy modes (calc_mems -> per-dimension entropy -> top_dims / k_dims z ->
    KV-cached greedy decode) and scripts/attention.py (forward_w_attn dump).  The checkpoint is the synthetic one."""
    from scripts import attention, parsers, sample
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    monkeypatch.chdir(tmp_path)
    cd, od, w = synth.load_vocab("pubchem")
    vae = TransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": 24, "BATCH_CHUNKS": 2},
                   name="c5", property_predictor=True, **synth.ARCHS["trans4x-256"])
    sd = synth.synth_state_dict(vae.model.state_dict(), 0)
    sd["encoder.z_means.weight"] = sd["encoder.z_means.weight"] * 0.05     # latent means inside calc_entropy's (-5, 5) histogram
    vae.model.load_state_dict(sd)
    vae.current_state["model_state_dict"] = vae.model.state_dict()
    vae.optimizer.sync_state_dict()
    vae.current_state["optimizer_state_dict"] = vae.optimizer.state_dict
    vae.save(vae.current_state, "cfg5", path="checkpoints")
    ckpt = "checkpoints/cfg5_c5.ckpt"
    # a loaded shell keeps the constructor's BATCH_SIZE of 500 (not an architecture parameter: reference :115-119), and
    # calc_mems only walks FULL batches -- so, as with the reference, the --mols file needs at least 500 molecules
    mols = [BASE[i % len(BASE)] + "C" * (i // len(BASE)) for i in range(500)]
    pd.DataFrame(mols, columns=["mol"]).to_csv("mols.txt", index=False)
    for mode, extra in (("high_entropy", []), ("k_high_entropy", ["--k", "5"])):
        out = "gen_%s.csv" % mode
        sample.sample(parsers.sample_parser().parse_args(
            ["--model", "transvae", "--model_ckpt", ckpt, "--mols", "mols.txt", "--sample_mode", mode, "--entropy_cutoff=-1e9",
             "--n_samples", "20", "--n_samples_per_batch", "10", "--save_path", out] + extra))
        got = pd.read_csv(out, keep_default_na=False)
        assert len(got) == 20 and got["mol"].map(lambda t: isinstance(t, str)).all()
    # the z of the two modes differ in what they leave at zero: k_dims draws on exactly k dimensions
    np.random.seed(0)
    z = vae.sample_from_memory(6, mode="k_dims", sample_dims=list(range(128)), k=5)
    assert int((z != 0).any(0).sum()) == 5
    attention.calc_attention(parsers.attn_parser().parse_args(
        ["--model", "transvae", "--model_ckpt", ckpt, "--mols", "mols.txt", "--n_samples", "24", "--batch_size", "12",
         "--batch_chunks", "2", "--save_path", "attn5"]))
    sa, xa = np.load("attn5_self_attn.npy"), np.load("attn5_src_attn.npy")
    assert sa.shape == (24, 4, 4, 127, 127) and xa.shape == (24, 3, 4, 126, 127)
    np.testing.assert_allclose(sa.sum(-1), 1.0, atol=1e-4)
    np.testing.assert_allclose(xa.sum(-1), 1.0, atol=1e-4)
