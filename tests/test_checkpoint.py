"""Checkpoint half of the Trainer mirror (reference trainer.py:396-579, optim.py:10-124,
generator.py:357-411) against a checkpoint WRITTEN BY THE REFERENCE's own ``Trainer.save`` and re-read by its own
``Trainer.resume`` (fixture ``tests/golden/ckpt_small``, made by ``oracle/make_golden_ckpt.py``).

CPU tests: file layout, path rules, strict loading, optimizer state, scheduler replay, step rounding, save round trip.
GPU test: after ``resume`` the HIP ExtraAdam continues exactly where the reference's optimizer continues."""
import shutil
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

from climategan_amd import fill
from climategan_amd.config import Opts
from climategan_amd.trainer import Trainer, _merge

FIX = Path(__file__).resolve().parent / "golden" / "ckpt_small"


def fixture_opts(output_path=None):
    o = Opts(yaml.safe_load((FIX / "opts.yaml").read_text()))
    o.output_path = str(output_path if output_path is not None else FIX)
    o.train.lambdas.G.p.vgg = 0          # no VGG19 needed for checkpoint tests
    return o


def make_trainer(output_path=None, device="cpu"):
    return Trainer(fixture_opts(output_path), device=device).setup(inference=False)


def seeded_grads(module, seed, device="cpu"):
    for i, p in enumerate(module.parameters()):
        if p.requires_grad:
            p.grad = torch.from_numpy(fill.uniform(tuple(p.shape), seed + i, -1e-2, 1e-2)).to(device)


def test_resume_reads_the_reference_checkpoint():
    ck = torch.load(FIX / "checkpoints" / "latest_ckpt.pth", map_location="cpu", weights_only=False)
    exp = np.load(FIX / "expected.npz")
    assert sorted(ck) == ["D", "G", "d_opt", "epoch", "g_opt", "step"]
    T = make_trainer()
    with pytest.warns(UserWarning):       # torch's "scheduler before optimizer" note: the reference triggers it too
        T.resume()
    # strict loading succeeded: the module trees have the reference's state-dict layout
    for name, mod in (("G", T.G), ("D", T.D)):
        sd = mod.state_dict()
        assert list(sd) == list(ck[name])
        for k, v in ck[name].items():
            assert torch.equal(sd[k], v), k
    # optimizer state: per-parameter step counts and both moments, in the reference's parameter order
    for opt, key in ((T.g_opt, "g_opt"), (T.d_opt, "d_opt")):
        ref_state = ck[key]["state"]
        params = [p for g in opt.param_groups for p in g["params"]]
        assert len(opt.param_groups) == len(ck[key]["param_groups"])
        assert [len(g["params"]) for g in opt.param_groups] == [len(g["params"]) for g in ck[key]["param_groups"]]
        assert set(ref_state) == {i for i, p in enumerate(params) if p.requires_grad}
        for i, st in ref_state.items():
            mine = opt.state[params[i]]
            assert int(mine["step"]) == int(st["step"]) == 2
            assert torch.equal(mine["exp_avg"], st["exp_avg"]) and torch.equal(mine["exp_avg_sq"], st["exp_avg_sq"])
    # counters and learning rates as the reference's own resume leaves them (odd step 13 -> 14)
    assert T.epoch == int(exp["epoch"][0]) == 7
    assert T.global_step == int(exp["step"][0]) == 14
    assert np.allclose([g["lr"] for g in T.g_opt.param_groups], exp["g_lr"], rtol=0, atol=0)
    assert np.allclose([g["lr"] for g in T.d_opt.param_groups], exp["d_lr"], rtol=0, atol=0)


def test_scheduler_replay_uses_the_pre_resume_epoch():
    """trainer.py:557-558 replays ``self.logger.epoch + 1`` scheduler steps BEFORE the epoch is restored; with the
    default StepLR (G: step 5, D: step 15, gamma 0.5) a trainer already at epoch 5 halves only the G rate."""
    T = make_trainer()
    T.epoch = 5
    with pytest.warns(UserWarning):
        T.resume()
    assert T.g_opt.param_groups[0]["lr"] == pytest.approx(0.00005 * 0.5)
    assert T.d_opt.param_groups[0]["lr"] == pytest.approx(0.00002)
    assert T.epoch == 7


def test_inference_resume_is_lenient_and_stops_at_G(capsys):
    o = fixture_opts()
    o.train.resume = True
    T = Trainer(o, device="cpu").setup(inference=True)      # setup itself resumes (trainer.py:735-736)
    ck = torch.load(FIX / "checkpoints" / "latest_ckpt.pth", map_location="cpu", weights_only=False)
    assert all(torch.equal(v, ck["G"][k]) for k, v in T.G.state_dict().items())
    assert T.D is None and not T.G.training
    # strict=False: unexpected / missing keys are reported, not fatal
    ck["G"]["painter.not_a_key"] = torch.zeros(1)
    del ck["G"]["painter.fc.bias"]
    tmp = Path(o.output_path)
    T2 = Trainer(o, device="cpu").setup(inference=False)
    T2._resolve_checkpoint = lambda: ck
    T2.resume(inference=True)
    out = capsys.readouterr().out
    assert "Missing keys" in out and "painter.fc.bias" in out and "painter.not_a_key" in out
    with pytest.raises(RuntimeError):
        T2.resume(inference=False)                            # training resume is strict
    assert tmp.exists()


def test_save_round_trip_and_epoch_files(tmp_path):
    T = make_trainer(tmp_path)
    with pytest.warns(UserWarning):
        T._resolve_checkpoint = lambda: torch.load(FIX / "checkpoints" / "latest_ckpt.pth", weights_only=False)
        T.resume()
    del T._resolve_checkpoint
    T.epoch, T.global_step = 50, 1000
    T.opts.train.min_save_epoch, T.opts.train.save_n_epochs = 28, 25
    T.save()
    files = sorted(p.name for p in (tmp_path / "checkpoints").iterdir())
    assert files == ["epoch_50_ckpt.pth", "latest_ckpt.pth"]
    T.epoch = 51
    T.save()                                               # 51 % 25 != 0: only latest is rewritten
    assert sorted(p.name for p in (tmp_path / "checkpoints").iterdir()) == files
    mine = torch.load(tmp_path / "checkpoints" / "latest_ckpt.pth", weights_only=False)
    ref = torch.load(FIX / "checkpoints" / "latest_ckpt.pth", weights_only=False)
    assert sorted(mine) == sorted(ref) and mine["epoch"] == 51 and mine["step"] == 1000
    for key in ("g_opt", "d_opt"):
        assert sorted(mine[key]) == sorted(ref[key]) == ["param_groups", "state"]
        assert sorted(mine[key]["param_groups"][0]) == sorted(ref[key]["param_groups"][0])
        assert mine[key]["param_groups"][0]["params"] == ref[key]["param_groups"][0]["params"]
        assert sorted(mine[key]["state"]) == sorted(ref[key]["state"])
        assert sorted(mine[key]["state"][0]) == sorted(ref[key]["state"][0])
    T2 = make_trainer(tmp_path)
    with pytest.warns(UserWarning):
        T2.resume()
    assert T2.epoch == 51 and T2.global_step == 1000
    for (k, a), (_, b) in zip(T.G.state_dict().items(), T2.G.state_dict().items()):
        assert torch.equal(a, b), k


def test_load_path_rules(tmp_path):
    """trainer.py:436-525: which file is read for which combination of tasks and load_paths."""
    def write(dirname, payload):
        d = tmp_path / dirname / "checkpoints"
        d.mkdir(parents=True)
        torch.save(payload, d / "latest_ckpt.pth")
        return tmp_path / dirname

    run_m = write("run_m", {"G": {"encoder.w": torch.ones(1)}, "epoch": 3, "step": 10})
    run_p = write("run_p", {"G": {"painter.w": torch.zeros(1)}, "epoch": 9, "step": 20})
    out = write("out", {"G": {"here": torch.ones(1)}, "epoch": 1, "step": 2})

    def resolve(tasks, **paths):
        o = fixture_opts(out)
        o.tasks = tasks
        for k in ("m", "p", "pm"):
            o.load_paths[k] = str(paths.get(k, "none"))
        T = Trainer.__new__(Trainer)
        T.opts, T.device = o, torch.device("cpu")
        return T._resolve_checkpoint()

    assert "here" in resolve(["m", "s", "d", "p"])["G"]                       # nothing given: output_path
    assert "painter.w" in resolve(["m", "p"], pm=run_p)["G"]                  # pm as a directory
    assert "encoder.w" in resolve(["m", "p"], pm=run_m / "checkpoints" / "latest_ckpt.pth")["G"]   # ... or a file
    merged = resolve(["m", "p"], m=run_m, p=run_p)                            # separate M and P runs: merged,
    assert sorted(merged["G"]) == ["encoder.w", "painter.w"]
    assert merged["epoch"] == 3 and merged["step"] == 10                      # the masker's scalars win (merge(m, p))
    with pytest.raises(ValueError, match="Cannot resume a P\\+M model"):
        resolve(["m", "p"], m=run_m, p=run_m)
    with pytest.raises(ValueError, match="received 2 values"):
        resolve(["p"], m=run_m, p=run_p)
    assert "encoder.w" in resolve(["m", "s", "d"], m=run_m)["G"]
    assert "painter.w" in resolve(["p"], p=run_p / "checkpoints" / "latest_ckpt.pth")["G"]
    with pytest.raises(AssertionError):
        resolve(["p"], m=run_m)                                               # masker path for a painter-only run
    with pytest.raises(AssertionError):
        resolve(["m", "p"], pm=tmp_path / "missing")
    assert _merge({"a": {"x": 1}, "b": 2}, {"a": {"y": 3}, "b": 0}) == {"a": {"x": 1, "y": 3}, "b": 2}


def test_load_val_painter(tmp_path, capsys):
    """generator.py:357-411: a masker-only generator borrows the Painter of another run (checkpoint FILE + the run's
    opts.yaml two levels up); failures are reported and answered with False."""
    run = tmp_path / "painter_run"
    shutil.copytree(FIX, run)
    o = fixture_opts(tmp_path)
    o.tasks = ["p"]
    from climategan_amd.generator import create_generator
    G = create_generator(o, device="cpu")
    for p in G.painter.parameters():
        torch.nn.init.constant_(p, 0.0)
    G.opts.val.val_painter = str(run / "checkpoints" / "latest_ckpt.pth")
    assert G.load_val_painter() is True
    ck = torch.load(run / "checkpoints" / "latest_ckpt.pth", weights_only=False)
    sd = G.painter.state_dict()
    assert all(torch.equal(sd[k.replace("painter.", "")], v) for k, v in ck["G"].items())
    assert not G.painter.training and not any(p.requires_grad for p in G.painter.parameters())
    G.opts.val.val_painter = str(run)                      # a directory is not accepted
    assert G.load_val_painter() is False
    assert "error (^) in load_val_painter" in capsys.readouterr().out


@pytest.mark.gpu
def test_hip_extra_adam_continues_like_the_reference_after_resume():
    """resume, then one extrapolation + one step on seeded gradients for G and D: parameters equal the ones the
    reference's optimizer produced from the same checkpoint (expected.npz)."""
    exp = np.load(FIX / "expected.npz")
    T = make_trainer(device="cuda")
    with pytest.warns(UserWarning):
        T.resume()
    for mod, opt, seed in ((T.G, T.g_opt, 7000), (T.D, T.d_opt, 8000)):
        seeded_grads(mod, seed, "cuda")
        opt.extrapolation()
        seeded_grads(mod, seed + 1000, "cuda")
        opt.step()
    worst = 0.0
    for name, mod in (("G.", T.G), ("D.", T.D)):
        for k, v in mod.state_dict().items():
            ref = exp[name + k]
            err = np.abs(v.cpu().numpy() - ref).max()
            worst = max(worst, err / max(np.abs(ref).max(), 1e-6))
    assert worst <= 2e-6, worst
