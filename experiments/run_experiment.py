"""Train / test entry point replacing the reference's ``experiments/*/exp*.py`` scripts (one runner, model chosen by its
paper name).  With ``--synthetic`` it needs no data: synthetic recordings whose targets are a fixed function of the input
(so that there is something to learn).

    python experiments/run_experiment.py --config tiny:SAUnet --synthetic 6 --epochs 3
    python experiments/run_experiment.py --config SAUnet:L --data DIR/hcqt --annot DIR/pitch --val 2382 --test 2628 --epochs 100
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from multipitch_architectures_amd import experiment  # noqa: E402
from multipitch_architectures_amd.synth import synth_file  # noqa: E402


def synthetic_recording(frames, seed):
    """noise as synth_file plus, for every active pitch p of the Bernoulli(0.04) target, energy at bin 3*p+1 shifted by
    each harmonic's offset (the pattern an HCQT shows for a tone) -- learnable from the centre frame"""
    inputs, targets = synth_file(frames=frames, seed=seed)
    t_idx, p_idx = np.nonzero(targets)
    for h, off in enumerate([-36, 0, 36, 57, 72, 83]):           # 3 bins per semitone, harmonics 0.5, 1, 2, 3, 4, 5
        b = 3 * p_idx + 1 + off
        ok = (b >= 0) & (b < inputs.shape[2])
        inputs[h, t_idx[ok], b[ok]] += 1.0 / (1 + h)
    return inputs, targets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny:SAUnet")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic recordings (train); 2 more for val/test")
    ap.add_argument("--data"), ap.add_argument("--annot")
    ap.add_argument("--val", nargs="*", default=[]), ap.add_argument("--test", nargs="*", default=[])
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--variant", default="Exp1", choices=sorted(experiment.VARIANTS),
                    help="Exp1: stride 50, whole epochs; Exp2 ('moresamples'/RETRAIN scripts): stride 20, epochs capped "
                         "after n_batches > 3800; Exp3 (Schubert splits): stride 10; Exp4 ('bigmix'): stride 35 + cap "
                         "(per-dataset strides: experiment.EXP4_STRIDES, passed per recording through the API)")
    ap.add_argument("--out", default=None, help="path of the best-model checkpoint (bare state_dict, as the scripts save it)")
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    torch.manual_seed(0)
    model, criterion, cfg = experiment.build(args.config)
    logging.info("Model: " + args.config + " (" + cfg["cls"] + "), " + str(sum(p.numel() for p in model.parameters())) + " parameters")
    if args.synthetic:
        train_files = [synthetic_recording(args.frames, 100 + k) for k in range(args.synthetic)]
        val_files, test_files, names = [synthetic_recording(args.frames, 7)], [synthetic_recording(args.frames // 2, 8)], ["synthetic-test"]
    else:
        all_files, all_names = experiment.load_musicnet_dir(args.data, args.annot, [""])
        pick = lambda vs: [f for f, n in zip(all_files, all_names) if any(v in n for v in vs)]
        val_files, test_files = pick(args.val), pick(args.test)
        names = [n for n in all_names if any(v in n for v in args.test)]
        train_files = [f for f, n in zip(all_files, all_names) if not any(v in n for v in args.val + args.test)]
    experiment.train(model, criterion, train_files, val_files, lr=cfg["lr"], max_epochs=args.epochs, path_trained_model=args.out,
                     variant=args.variant)
    if args.out:
        model.load_state_dict(torch.load(args.out))
    experiment.test(model, test_files, names)


if __name__ == "__main__":
    main()
