"""CPU: CLI plumbing -- flag parsing in the reference's style, folder layout, checkpoint naming, uint16 .npy code files."""
import json
import os

import numpy as np
import pytest
import torch

from synthanatomy_amd.utils import general as G


def test_flag_parsing_literals_aliases_and_errors():
    d = dict(a=1, roi=((0, 1),), name="x", flag=False, req=G.REQUIRED, n_embd=8)
    cfg = G.parse_flags(["run", "--a=3", "--roi=((16,176),(16,240),(96,256))", "--name=foo", "--flag", "--req=/p/", "--n_embed=512"], d, {"n_embed": "n_embd"})
    assert cfg == dict(a=3, roi=((16, 176), (16, 240), (96, 256)), name="foo", flag=True, req="/p/", n_embd=512)
    with pytest.raises(SystemExit):
        G.parse_flags(["--nope=1"], d)
    with pytest.raises(SystemExit):
        G.parse_flags(["--a=1"], d)  # required flag missing


def test_folder_layout_checkpoints_and_npy(tmp_path):
    cfg = dict(project_directory=str(tmp_path) + "/", experiment_name="exp", network="baseline_vqvae", starting_epoch=0)
    G.create_folder_structure(cfg)
    base = tmp_path / "exp" / "baseline_vqvae"
    assert all((base / s).is_dir() for s in ("checkpoints", "logs", "outputs", "caching")) and cfg["starting_epoch"] == 0
    net = torch.nn.Linear(2, 2)
    p0 = G.save_checkpoint(cfg, 1, {"network": net, "trainer": {"iteration": 3}})
    p1 = G.save_checkpoint(cfg, 2, {"network": torch.nn.DataParallel(net), "trainer": {"iteration": 6}, "d_network": None})
    assert os.path.basename(p1) == "checkpoint_epoch=2.pt" and not os.path.exists(p0)  # n_saved = 1
    obj = torch.load(p1, weights_only=False)
    assert set(obj) == {"network", "trainer"} and set(obj["network"]) == {"weight", "bias"}     # DDP-style wrappers are unwrapped
    assert G.latest_checkpoint(cfg["checkpoint_directory"]) == (p1, 2)
    cfg2 = dict(cfg, starting_epoch=0)
    G.create_folder_structure(cfg2)
    assert cfg2["starting_epoch"] == -1  # non-empty checkpoint dir -> resume
    torch.save({"module." + k: v for k, v in net.state_dict().items()}, tmp_path / "m.pt")
    res, _ = G.load_network_state(torch.nn.Linear(2, 2), str(tmp_path / "m.pt"))
    assert not res.missing_keys and not res.unexpected_keys
    path = G.save_npy(np.arange(24).reshape(2, 3, 4), cfg["outputs_directory"], "/data/sub-01_T1w.nii.gz", "quantization_0")
    assert path.endswith("outputs/sub-01_T1w/sub-01_T1w_quantization_0.npy") and np.load(path).dtype == np.uint16
    assert G.list_inputs("synthetic:3") == ["synthetic_0000", "synthetic_0001", "synthetic_0002"]
    assert G.list_inputs(cfg["outputs_directory"]) == [path]


def test_cli_rejects_unknown_modes_before_touching_the_gpu(tmp_path):
    import run_transformer
    import run_vqvae
    base = ["--training_subjects=synthetic:1", "--validation_subjects=synthetic:1", f"--project_directory={tmp_path}/", "--experiment_name=e"]
    with pytest.raises(ValueError):
        run_vqvae.run(base + ["--mode=bogus"])
    with pytest.raises(ValueError):
        run_transformer.run(base + ["--mode=bogus"])


def test_checkpoint_selection_rules(tmp_path):
    """reference src/utils/general.py:75-168: training resumes from the newest / the requested epoch; evaluation takes the requested epoch,
    the newest ("recent") or the single key-metric checkpoint ("best")."""
    cfg = dict(project_directory=str(tmp_path) + "/", experiment_name="exp", network="performer", starting_epoch=0, mode="training")
    G.create_folder_structure(cfg)
    assert G.check_for_checkpoints(cfg) is None and cfg["starting_epoch"] == 0
    net = torch.nn.Linear(2, 2)
    p3 = G.save_checkpoint(cfg, 3, {"network": net})
    b1 = G.save_checkpoint(cfg, 1, {"network": net}, key_metric=-0.5)
    assert G.save_checkpoint(cfg, 2, {"network": net}, key_metric=-0.7) is None and os.path.exists(b1)      # worse: the best one stays
    b2 = G.save_checkpoint(cfg, 3, {"network": net}, key_metric=-0.25)
    assert os.path.basename(b2) == "checkpoint_key_metric=-0.2500.pt" and not os.path.exists(b1) and os.path.exists(p3)
    tr = dict(cfg, starting_epoch=-1)
    assert G.check_for_checkpoints(tr) == p3 and tr["starting_epoch"] == 3
    with pytest.raises(AssertionError):
        G.check_for_checkpoints(dict(cfg, starting_epoch=7))
    assert G.check_for_checkpoints(dict(cfg, mode="extracting", starting_epoch=0, evaluation_checkpoint="recent")) == p3
    assert G.check_for_checkpoints(dict(cfg, mode="extracting", starting_epoch=0, evaluation_checkpoint="best")) == b2
    assert G.check_for_checkpoints(dict(cfg, mode="inference", starting_epoch=3, evaluation_checkpoint="best")) == p3
    with pytest.raises(ValueError):
        G.load_checkpoint(p3, {"optimizer": net})     # ignite's CheckpointLoader: a requested key that is absent is an error


def test_best_checkpoint_is_chosen_by_the_unrounded_metric(tmp_path):
    """key metrics of run_vqvae are -MSE, i.e. 1e-3 .. 1e-5: the comparison must not go through the '%.4f' file name (ignite keeps the score in
    memory), neither within one process nor after a restart."""
    cfg = dict(project_directory=str(tmp_path) + "/", experiment_name="exp", network="baseline_vqvae", starting_epoch=0, mode="training")
    G.create_folder_structure(cfg)
    net = torch.nn.Linear(2, 2)
    a = G.save_checkpoint(cfg, 1, {"network": net}, key_metric=-0.00054)
    assert os.path.basename(a) == "checkpoint_key_metric=-0.0005.pt"
    b = G.save_checkpoint(cfg, 2, {"network": net}, key_metric=-0.00052)          # better, same rounded name
    assert b is not None and sorted(torch.load(b, weights_only=False)) == ["network"]          # exactly the reference's to_save keys
    assert json.load(open(os.path.join(cfg["checkpoint_directory"], G._SIDECAR))) == {"file": os.path.basename(b), "score": -0.00052}
    assert G.save_checkpoint(cfg, 3, {"network": net}, key_metric=-0.00053) is None
    c = G.save_checkpoint(cfg, 4, {"network": net}, key_metric=-3e-5)
    assert os.path.basename(c) == "checkpoint_key_metric=-0.0000.pt" and not os.path.exists(b)
    G._BEST_SCORE.clear()                                                          # a restarted process: the score comes back from the sidecar file
    assert G.save_checkpoint(cfg, 5, {"network": net}, key_metric=-4e-5) is None
    d = G.save_checkpoint(cfg, 6, {"network": net}, key_metric=-1e-5)
    assert d is not None and json.load(open(os.path.join(cfg["checkpoint_directory"], G._SIDECAR)))["score"] == -1e-5
    assert sorted(os.listdir(cfg["checkpoint_directory"])) == sorted([os.path.basename(d), G._SIDECAR])
    G.load_checkpoint(d, {"network": net})
    # a lost sidecar / a reference-written file: the score is parsed back from the name; a foreign name is skipped, not an AttributeError
    os.remove(os.path.join(cfg["checkpoint_directory"], G._SIDECAR))
    G._BEST_SCORE.clear()
    assert G._best_key_metric(cfg["checkpoint_directory"], [d, os.path.join(cfg["checkpoint_directory"], "checkpoint_key_metric_foreign.pt")]) == 0.0


def test_resume_uses_the_checkpoints_epoch_length():
    """10 epochs of 100 iterations saved on one GPU, resumed on two (epoch length 50): the run continues at epoch 10, not 20."""
    from synthanatomy_amd.runtime.optim import TrainerState
    st = TrainerState(epoch_length=50, max_epochs=30)
    st.load_state_dict({"iteration": 1000, "epoch_length": 100, "max_epochs": 20})
    assert st.epoch == 10
    assert st.rebase(50, 30) == 10
    assert st.epoch == 10 and st.iteration == 500 and st.state_dict() == {"iteration": 500, "epoch_length": 50, "max_epochs": 30}


def test_distributed_sampler_sharding():
    """Every rank gets the same number of samples (wrap-around padding), one permutation per epoch shared by all ranks."""
    for n, world in [(5, 2), (7, 4), (8, 4), (1, 2)]:
        shards = [G.shard_for_rank(n, r, world, epoch=3, seed=4) for r in range(world)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == (n + world - 1) // world
        assert set(i for s in shards for i in s) == set(range(n))
    assert G.shard_for_rank(6, 0, 2, epoch=0, seed=4) != G.shard_for_rank(6, 0, 2, epoch=1, seed=4)          # reshuffled every epoch
    import torch.utils.data as tud
    ds = list(range(11))
    for r in range(3):
        sam = tud.DistributedSampler(ds, num_replicas=3, rank=r, shuffle=True, seed=4)
        sam.set_epoch(2)
        assert list(sam) == G.shard_for_rank(11, r, 3, epoch=2, seed=4)      # identical to torch's sampler
    assert [G.shard_for_rank(5, r, 2, shuffle=False, pad=False) for r in range(2)] == [[0, 2, 4], [1, 3]]


def test_optimizer_scheduler_and_trainer_state_use_the_reference_layout():
    """The ``optimizer`` / ``lr_scheduler`` / ``trainer`` entries of a checkpoint have the layout torch.optim.Adam / ExponentialLR / ignite write,
    so the reference's checkpoints load here and ours load there (run_vqvae.py:312-345)."""
    from synthanatomy_amd.runtime.optim import ExponentialLR, FlatParams, FusedAdam, TrainerState
    torch.manual_seed(0)
    mod = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Embedding(4, 2), torch.nn.Linear(5, 1))
    mod[1].weight.requires_grad_(False)                      # like the frozen codebook: numbered by Adam, never updated
    ref_opt = torch.optim.Adam(mod.parameters(), lr=2e-3)
    ref_sch = torch.optim.lr_scheduler.ExponentialLR(ref_opt, gamma=0.9)
    for _ in range(3):
        ref_opt.zero_grad()
        mod[2](mod[0](torch.randn(4, 3))).sum().backward()
        ref_opt.step()
        ref_sch.step()
    flat = FlatParams(mod.parameters())
    opt = FusedAdam(flat, lr=1.0)
    sch = ExponentialLR(opt, gamma=0.5)
    opt.load_state_dict(ref_opt.state_dict())
    sch.load_state_dict(ref_sch.state_dict())
    assert opt.step_count == 3 and abs(opt.lr - 2e-3 * 0.9 ** 3) < 1e-12 and sch.gamma == 0.9 and sch.last_epoch == 3
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and sorted(sd["state"]) == [0, 1, 3, 4] and sd["param_groups"][0]["params"] == [0, 1, 2, 3, 4]
    for k, ent in ref_opt.state_dict()["state"].items():
        assert torch.equal(sd["state"][k]["exp_avg"], ent["exp_avg"]) and torch.equal(sd["state"][k]["exp_avg_sq"], ent["exp_avg_sq"])
        assert float(sd["state"][k]["step"]) == 3.0
    fresh = torch.optim.Adam(mod.parameters(), lr=1.0)
    fresh.load_state_dict(sd)                                 # and torch accepts ours
    assert fresh.state_dict()["param_groups"][0]["lr"] == opt.lr
    assert set(sch.state_dict()) >= {"gamma", "base_lrs", "last_epoch", "_last_lr", "_step_count"}
    st = TrainerState(epoch_length=7, max_epochs=10)
    st.load_state_dict({"iteration": 21, "epoch_length": 7, "max_epochs": 10})
    assert st.epoch == 3 and st.state_dict() == {"iteration": 21, "epoch_length": 7, "max_epochs": 10}
