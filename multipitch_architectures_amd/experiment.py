"""One runner for the train / validate / test flow that every script under the reference's ``experiments/`` repeats
(e.g. ``Exp1_SectionIV-B/exp180d_musicnet_unet_extremelylarge_doubleselfattn.py:252-505``), parameterised by the paper
name of the model (``configs.CONFIGS``) instead of 111 copies.  The data pipeline, the model, the loss, the optimiser and
the evaluation measures all run on the GPU (``data_loaders``, ``nn_models``, ``losses``, ``optim``, ``metrics``); only
the epoch-level control flow -- ReduceLROnPlateau, early stopping, best-checkpoint saving, the log lines -- is host code.

Reproduced as the scripts have it:
* batch 25 / 50 / 50, context 75, stride 50 (train/val) and 1 (test), compression 10, the four augmentations on the
  training sets only (:38-65); AdamW(1e-3, (0.9, 0.999), 1e-8, wd 0.01), ReduceLROnPlateau(0.5, patience 5,
  threshold 1e-4 rel, min_lr 1e-6), early stopping (min, 1e-5, patience 12) (:95-145);
* validation runs in *train mode* -- the scripts never call ``model.eval()`` before it (:339-350), so BatchNorm uses
  batch statistics there and keeps updating its running estimates; ``model.eval()`` only precedes testing (:398);
* the log lines (:353-378, :440-505): ``Epoch #k finished. Train Loss: …, Val Loss: … with lr: …``,
  ``  .... model of epoch #k saved.``, ``file … tested. Cosine sim: …``, ``Mean <measure>:   …``,
  ``Framewise <measure>:   …``.
Not reproduced: the mir_eval columns, the CSV/prediction dumps, the cluster paths.
"""
import logging
import os

import numpy as np
import torch

from . import nn_models, ops
from .configs import CONFIGS
from .data_loaders import ContextLoader, dataset_context, dataset_context_segm
from .losses import BCELoss, PolyphonyLoss
from .metrics import MEASURES, calculate_eval_measures
from .metrics.eval_metrics import aggregate_files
from .metrics.monitoring import early_stopping
from .optim import AdamW
from .step import TrainStep

TRAIN_DATASET_PARAMS = {"context": 75, "stride": 50, "compression": 10, "aug:transpsemitones": 5, "aug:randomeq": 20,
                        "aug:noisestd": 1e-4, "aug:tuning": True}
VAL_DATASET_PARAMS = {"context": 75, "stride": 50, "compression": 10}
TEST_DATASET_PARAMS = {"context": 75, "stride": 1, "compression": 10}
EVAL_THRESH = 0.4

# What differs between the experiment families (everything else -- batch sizes, optimiser, scheduler, early stopping,
# measures -- is identical in all 111 scripts):
#   Exp1 (Section IV-B)        stride 50 for train / val, whole epochs                       exp180d...py:38-50
#   Exp2 (Section IV-C, "moresamples" / RETRAIN*)  stride 20 and an epoch ends after the batch that makes
#        n_batches > 3800, i.e. after at most 3801 batches                                   RETRAIN_exp180d...py:38-50,337-338
#   Exp3 (Section IV-D, Schubert Winterreise splits)  stride 10, whole epochs                exp200a...py:38-50
#   Exp4 (Section IV-E, "bigmix": five training datasets concatenated)  a stride per dataset and the 3800-batch cap
#        (exp210d...py:38-50, 310-311, 359-360, 405, 437-438, 535): pass (inputs, targets, stride) triples as files --
#        a third element overrides the variant's stride for that recording; EXP4_STRIDES holds the scripts' values
VARIANTS = {"Exp1": {"stride": 50, "max_batches": None}, "Exp2": {"stride": 20, "max_batches": 3800},
            "Exp3": {"stride": 10, "max_batches": None}, "Exp4": {"stride": 35, "max_batches": 3800}}
# dataset -> (train stride, validation stride; None: the dataset has no validation part)
EXP4_STRIDES = {"MusicNet": (35, 35), "Schubert_Winterreise": (6, 4), "Bach10": (1, 1), "PHENICX-Anechoic": (2, None),
                "ChoralSingingDataset": (4, 4)}


def build(config, device="cuda:0"):
    cfg = CONFIGS[config]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(device)
    is_punet = cfg["cls"].endswith("polyphony_classif_softmax")
    pl = PolyphonyLoss() if is_punet else None
    bce = BCELoss(reduction="mean")
    criterion = (lambda res, y: pl(res[0], res[1], y)) if is_punet else (lambda res, y: bce(res, y))
    return model, criterion, cfg


def train(model, criterion, train_files, val_files, lr=1e-3, max_epochs=100, batch_sizes=(25, 50), seed=0,
          path_trained_model=None, log=logging.info, rank=0, world=1, averager=None, variant="Exp1", stride=None,
          max_batches=None, use_graph=True):
    """train_files / val_files: lists of (inputs (6,T,216), targets (T,n_out)[, stride]) tuples.  Returns the per-epoch history.
    ``variant`` picks the experiment family's stride and per-epoch batch cap (``VARIANTS``); ``stride`` /
    ``max_batches`` override it.  ``use_graph``: the loop body (forward, loss, backward, AdamW) is captured once as a HIP
    graph and replayed for every full-size batch (``step.TrainStep``); the last, smaller batch of an epoch launches kernel
    by kernel.  Data-parallel runs (``averager``) replay one graph per gradient bucket with the bucket's all-reduce launched
    behind it, overlapping the rest of backward (step.py)."""
    var = VARIANTS[variant]
    stride = var["stride"] if stride is None else stride
    max_batches = var["max_batches"] if max_batches is None else max_batches
    own = lambda f: f[2] if len(f) > 2 and f[2] is not None else stride         # per-recording stride (Exp4)
    train_sets = [dataset_context(f[0], f[1], dict(TRAIN_DATASET_PARAMS, stride=own(f)), seed=seed + k)
                  for k, f in enumerate(train_files)]
    val_sets = [dataset_context(f[0], f[1], dict(VAL_DATASET_PARAMS, stride=own(f))) for f in val_files]
    train_loader = ContextLoader(train_sets, batch_sizes[0], shuffle=True, seed=seed, rank=rank, world=world)
    val_loader = ContextLoader(val_sets, batch_sizes[1], shuffle=False)
    optimizer = AdamW(model.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, amsgrad=False)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=5, threshold=1e-4,
                                                           threshold_mode="rel", cooldown=0, eps=1e-8, min_lr=1e-6)
    es = early_stopping(mode="min", min_delta=1e-5, patience=12, percentage=False)
    log("\n \n ###################### START TRAINING ###################### \n")
    history = []
    model.train()
    train_step = TrainStep(model, criterion, optimizer, averager=averager, use_graph=use_graph)
    for epoch in range(max_epochs):
        accum_loss, n_batches = 0.0, 0
        for local_batch, local_labels in train_loader:
            loss = train_step(local_batch, local_labels)     # y_pred = model(x); loss; zero_grad; backward; step
            accum_loss += loss.item()
            n_batches += 1
            if max_batches is not None and n_batches > max_batches:       # RETRAIN_exp180d...py:337-338
                break
        train_loss = accum_loss / max(n_batches, 1)
        accum_val_loss, n_val = 0.0, 0
        with torch.no_grad():          # still train mode, as in the scripts: BN batch statistics, running stats move
            for local_batch, local_labels in val_loader:
                accum_val_loss += criterion(model(local_batch), local_labels).item()
                n_val += 1
        ops.rng_advance()              # validation ran in train mode (dropout active): move the dropout stream past it
        val_loss = accum_val_loss / max(n_val, 1)
        log("Epoch #" + str(epoch) + " finished. Train Loss: " + "{:.4f}".format(train_loss) + ", Val Loss: " +
            "{:.4f}".format(val_loss) + " with lr: " + "{:.5f}".format(optimizer.param_groups[0]["lr"]))
        scheduler.step(val_loss)
        history.append((train_loss, val_loss))
        if epoch == 0 or es.curr_is_better(val_loss):
            if path_trained_model:
                torch.save(model.state_dict(), path_trained_model)
            log("  .... model of epoch " + ("0" if epoch == 0 else "#" + str(epoch)) + " saved.")
        if es.step(val_loss):
            break
    if path_trained_model:
        log(" ### trained model saved in " + path_trained_model + " \n")
    return history


@torch.no_grad()
def predict_file(model, inputs, targets, batch_size=50, segment=None):
    """frame-wise predictions of one recording, (T, n_out), on the device.

    ``segment=None`` -- the reference's test loop (exp180d...py:420-440): pad half a context on both sides, one
    75-frame patch per output frame (stride 1), batches of 50.

    ``segment=L`` -- opt-in fast path: the padded recording is cut into windows of L + 74 frames at
    stride L (the ``dataset_context_segm`` shape, hcqt_datasets.py:144-289) and every forward pass writes L output
    frames: L x less convolution work per frame for the large layers.  The models are fully convolutional in time
    (T >= 75 in, T - 74 out), so this runs the same kernels; it is *not* bit-identical to the per-patch loop for any
    family: a 75-frame patch zero-pads the 15x15 / 3x3 convolutions and the time pools at *its own* borders, a long
    window only at the window's (and the U-Nets additionally align their 2x2 pooling grid to the window start).  What
    holds exactly: ``segment=1`` is the per-patch loop, and window k of the result is ``model(window k)``.  The
    deviation from the per-patch predictions is reported by ``bench.py`` (``segment_inference``).  Models with the
    batch-axis attention (SAUnet / SAUSnet) mix the windows of a batch exactly as they mix the patches of a batch."""
    half = TEST_DATASET_PARAMS["context"] // 2
    inputs = np.pad(np.asarray(inputs), ((0, 0), (half, half + 1), (0, 0)))
    targets_p = np.pad(np.asarray(targets), ((half, half + 1), (0, 0)))
    first = lambda res: res[0] if isinstance(res, tuple) else res
    if segment is None:
        ds = dataset_context(inputs, targets_p, dict(TEST_DATASET_PARAMS))
        preds = []
        for X, _ in ContextLoader([ds], batch_size, shuffle=False):
            preds.append(first(model(X)).squeeze(2).squeeze(1))
        return torch.cat(preds)                      # (T, n_out), stays on the device
    L = int(segment)
    if L < 1:
        raise ValueError("segment must be a positive number of frames")
    T = inputs.shape[1] - 2 * half - 1               # frames of the recording = frames to predict
    n_full, rem = divmod(T, L)
    preds = []
    if n_full:
        ds = dataset_context_segm(inputs, targets_p, dict(TEST_DATASET_PARAMS, seglength=L, stride=L))
        per_batch = max(1, (batch_size * 75) // (L + 74))     # about as many input frames per forward as the patch loop
        for i in range(0, n_full, per_batch):
            X, _ = ds.batch(list(range(i, min(i + per_batch, n_full))))
            y = first(model(X))                      # (b, 1, L, n_out)
            preds.append(y.squeeze(1).reshape(-1, y.shape[-1]))
    if rem:                                          # the last, shorter window: rem + 74 frames -> rem frames
        tail = inputs[:, n_full * L:, :]
        ds = dataset_context_segm(tail, targets_p[n_full * L:], dict(TEST_DATASET_PARAMS, seglength=rem, stride=rem))
        X, _ = ds.batch([0])
        y = first(model(X))
        preds.append(y.squeeze(1).reshape(-1, y.shape[-1]))
    return torch.cat(preds)


def test(model, test_files, names=None, measures=MEASURES, log=logging.info, segment=None):
    log("\n \n ###################### START TESTING ###################### \n")
    model.eval()
    per_file, n_frames = [], []
    for k, (inputs, targets) in enumerate(test_files):
        pred = predict_file(model, inputs, targets, segment=segment)
        targ = torch.as_tensor(np.asarray(targets), dtype=torch.float32)
        assert tuple(pred.shape) == tuple(targ.shape), "Shape mismatch! Target shape: " + str(tuple(targ.shape)) + \
            ", Pred. shape: " + str(tuple(pred.shape))
        ev = calculate_eval_measures(targ, pred, measures=list(measures), threshold=EVAL_THRESH)
        per_file.append([ev[m] for m in measures])
        n_frames.append(targ.shape[0])
        log("file " + str(names[k] if names else k) + " tested. Cosine sim: " + str(ev.get("cosine_sim")))
    log("### Testing done. ################################################ \n")
    mean, framewise = aggregate_files(per_file, n_frames)
    for m, v in zip(measures, mean):
        log("Mean " + m + ":   " + str(v))
    log("\n")
    for m, v in zip(measures, framewise):
        log("Framewise " + m + ":   " + str(v))
    return dict(zip(measures, mean)), dict(zip(measures, framewise))


def load_musicnet_dir(path_data, path_annot, versions, num_output_bins=72, min_pitch=24):
    """recordings whose file name contains one of ``versions``, laid out as the scripts expect (:262-265): HCQT .npy of
    shape (bins, frames, harmonics) and annotation .npy of shape (pitches, frames)."""
    files, names = [], []
    for fn in sorted(os.listdir(path_data)):
        if any(v in fn for v in versions):
            inputs = np.transpose(np.load(os.path.join(path_data, fn)), (2, 1, 0))
            targets = np.load(os.path.join(path_annot, fn)).T
            if num_output_bins != 12:
                targets = targets[:, min_pitch:(min_pitch + num_output_bins)]
            files.append((inputs, targets))
            names.append(fn)
    return files, names
