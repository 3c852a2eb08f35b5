"""A/B of a real (small) training run in the two arithmetics: experiments/run_experiment.py's synthetic task, same seeds, same
data order, exact fp32 against ops.set_conv_precision("bf16x3").  Per epoch: train loss, validation loss; at the end the test
measures of the reference's test flow (threshold 0.4).  Writes gpurun_out/r04_bf16x3_train_ab.json.
usage: python scratch/bfx_train_ab.py [config ...]   (default: tiny:SAUnet tiny:DRCNN)"""
import json, logging, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "experiments"))
import torch
from multipitch_architectures_amd import experiment, ops
from run_experiment import synthetic_recording

EPOCHS = int(os.environ.get("AB_EPOCHS", "12")); NREC = int(os.environ.get("AB_RECORDINGS", "6")); FRAMES = int(os.environ.get("AB_FRAMES", "1500"))


def run(config, precision, seed=1234):
    ops.set_conv_precision(precision)
    ops.manual_seed(seed)                 # dropout stream; weights, data and batch order are the same in every run
    torch.manual_seed(0)
    model, criterion, cfg = experiment.build(config)
    train_files = [synthetic_recording(FRAMES, 100 + k) for k in range(NREC)]
    val_files, test_files = [synthetic_recording(FRAMES, 7)], [synthetic_recording(FRAMES // 2, 8)]
    t0 = time.time()
    ckpt = os.path.join("/tmp", f"ab_{config.replace(':', '_')}_{precision}_{seed}.pt")
    hist = experiment.train(model, criterion, train_files, val_files, lr=cfg["lr"], max_epochs=EPOCHS, log=lambda *_: None,
                            path_trained_model=ckpt)
    model.load_state_dict(torch.load(ckpt))          # the best-validation model, as the scripts test it (exp126a...py:383-391)
    mean, _ = experiment.test(model, test_files, ["synthetic-test"], log=lambda *_: None)
    ops.set_conv_precision("f32")
    return {"precision": precision, "dropout_seed": seed, "epochs": len(hist), "train_loss": [h[0] for h in hist], "val_loss": [h[1] for h in hist],
            "test": {k: float(v) for k, v in mean.items()}, "seconds": time.time() - t0}


if __name__ == "__main__":
    logging.basicConfig(level=logging.WARNING)
    out = {"task": f"experiments/run_experiment.py --synthetic {NREC} --frames {FRAMES} --epochs {EPOCHS}", "runs": {}}
    for config in (sys.argv[1:] or ["tiny:SAUnet", "tiny:DRCNN"]):
        seeds = [1234, 4321, 777, 2024, 99][:int(os.environ.get("AB_SEEDS", "5"))]
        runs = {prec: [run(config, prec, seed=sd) for sd in seeds] for prec in ("f32", "bf16x3")}
        F = lambda r: r["test"]["f_measure"]
        import statistics as st
        summ = {prec: {"f_measure": [F(r) for r in rs], "mean": st.mean(F(r) for r in rs),
                       "stdev": st.stdev(F(r) for r in rs) if len(rs) > 1 else 0.0,
                       "best_val_loss": [min(r["val_loss"]) for r in rs]} for prec, rs in runs.items()}
        out["runs"][config] = {"dropout_seeds": seeds, "summary": summ, "runs": runs,
                               "mean_f_measure_diff_pp": 100.0 * abs(summ["f32"]["mean"] - summ["bf16x3"]["mean"])}
        print(config, "F f32", [round(v, 3) for v in summ["f32"]["f_measure"]], "mean", round(summ["f32"]["mean"], 4), "+-",
              round(summ["f32"]["stdev"], 4), "| bf16x3", [round(v, 3) for v in summ["bf16x3"]["f_measure"]], "mean",
              round(summ["bf16x3"]["mean"], 4), "+-", round(summ["bf16x3"]["stdev"], 4), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_bf16x3_train_ab.json"), "w"), indent=1)
