"""Host-side mirror of the reference's train.py step functions (train.py:22-44) plus the
data-parallel wrapper the reference lacks."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib, losses, parallel
from .models import SeldNet


class Adam:
    """tf.keras.optimizers.Adam(learning_rate) (train.py:311): the slots live in the HIP ctx."""

    def __init__(self, learning_rate: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon


def _cfg(doa_loss, loss_weight: Sequence[float], sed_grad_scale: float = 1.0, mmse_den: float = 0.0) -> _lib.LossCfg:
    if not isinstance(doa_loss, (losses._MSE, losses._MMSE)):
        raise ValueError("doa_loss must be seld_amd.losses.MSE, .MAE, .MSLE or .MMSE")
    return _lib.LossCfg(doa_loss.code, float(loss_weight[0]), float(loss_weight[1]), float(sed_grad_scale), float(mmse_den))


def _labels(model: SeldNet, y, B: int):
    ys = torch.as_tensor(y[0], dtype=torch.float32, device=model._dev).contiguous()
    yd = torch.as_tensor(y[1], dtype=torch.float32, device=model._dev).contiguous()
    if tuple(ys.shape) != (B, model.S, model.n_classes) or tuple(yd.shape) != (B, model.S, 3 * model.n_classes):
        raise ValueError(f"label shapes {tuple(ys.shape)}, {tuple(yd.shape)} do not match the model output")
    return ys, yd


def _loss_outputs(model: SeldNet, doa_loss, B: int):
    sloss = torch.empty((), dtype=torch.float32, device=model._dev)
    dshape = (B, model.S) if isinstance(doa_loss, losses._MSE) else ()
    dloss = torch.empty(dshape, dtype=torch.float32, device=model._dev)
    return sloss, dloss


def trainstep(model: SeldNet, x, y, sed_loss, doa_loss, loss_weight, optimizer: Adam, agc: bool = False,
              process_group=None, allreduce: bool = True):
    """reference train.trainstep (train.py:22-36) -> (y_p, sloss, dloss).

    With torch.distributed initialised (one process per GPU) the flat gradient buffer is summed over
    ranks with one RCCL all-reduce between backward and Adam; BatchNorm statistics stay per replica."""
    if not isinstance(sed_loss, losses.BinaryCrossentropy):
        raise ValueError("sed_loss must be seld_amd.losses.BinaryCrossentropy()")
    if not isinstance(model, SeldNet):       # modules.ComposedSeldNet (FIRST = mother_block / mother_stage): single process
        if parallel.world_size(process_group) > 1:
            raise ValueError("composed models are not data-parallel")
        return model.train_step(x, y, _cfg(doa_loss, loss_weight), optimizer, agc)
    x = model._prep(x)
    B = x.shape[0]
    ys, yd = _labels(model, y, B)
    sed, doa = model._outputs(B)
    sloss, dloss = _loss_outputs(model, doa_loss, B)
    world = parallel.world_size(process_group)
    is_mmse = isinstance(doa_loss, losses._MMSE)
    lib_dp = getattr(model, "_lib_dp", False)        # parallel.init_library_dp: the library owns the RCCL communicator
    if world > 1 and not lib_dp and not getattr(model, "_dp_seeded", False):
        # torch.distributed path: every replica needs its own dropout key too (init_library_dp sets it on the library path), or all
        # ranks would drop the same elements of their different clips
        model.set_option("dropout_seed", 0x5e1d + torch.distributed.get_rank(process_group))
        model._dp_seeded = True
    dent = None
    if lib_dp:
        pass            # the mask count is all-reduced on the device inside seld_train_fwd_bwd (cfg.mmse_den = 0)
    elif world > 1 and is_mmse:
        # scalar objective: BCE is a mean over the GLOBAL batch, MMSE divides by the GLOBAL sum(mask)
        dent = torch.empty(1, dtype=torch.float32, device=model._dev)
        _lib.check(model.lib.seld_mmse_den(model.ctx, yd.data_ptr(), dent.data_ptr()), model.ctx)
    if lib_dp:
        sed_scale, den = (1.0 / int(model.lib.seld_dp_world(model.ctx)) if is_mmse else 1.0), 0.0
    else:
        sed_scale, den = parallel.loss_scaling(is_mmse, dent, process_group)
    cfg = _cfg(doa_loss, loss_weight, sed_scale, den)
    _lib.check(model.lib.seld_train_fwd_bwd(model.ctx, x.data_ptr(), ys.data_ptr(), yd.data_ptr(), C.byref(cfg),
                                            sed.data_ptr(), doa.data_ptr(), sloss.data_ptr(), dloss.data_ptr()), model.ctx)
    if (world > 1 or lib_dp) and allreduce:      # allreduce=False: timing aid only (bench.py measures the exposed communication time with it)
        parallel.allreduce_gradients(model.grad_tensor(), process_group, model)
    _lib.check(model.lib.seld_adam_step(model.ctx, optimizer.learning_rate, optimizer.beta_1, optimizer.beta_2,
                                        optimizer.epsilon, int(bool(agc))), model.ctx)
    return [sed, doa], sloss, dloss


def teststep(model: SeldNet, x, y, sed_loss, doa_loss):
    """reference train.teststep (train.py:39-44) -> (y_p, sloss, dloss)."""
    if not isinstance(model, SeldNet):       # modules.ComposedSeldNet
        return model.test_step(x, y, _cfg(doa_loss, (1.0, 1.0)))
    x = model._prep(x)
    B = x.shape[0]
    ys, yd = _labels(model, y, B)
    sed, doa = model._outputs(B)
    sloss, dloss = _loss_outputs(model, doa_loss, B)
    cfg = _cfg(doa_loss, (1.0, 1.0))
    _lib.check(model.lib.seld_test_step(model.ctx, x.data_ptr(), ys.data_ptr(), yd.data_ptr(), C.byref(cfg),
                                        sed.data_ptr(), doa.data_ptr(), sloss.data_ptr(), dloss.data_ptr()), model.ctx)
    return [sed, doa], sloss, dloss


# ---------------------------------------------------------------------------------------------------
def get_dataset(config, mode: str = 'train', device=None):
    """reference train.get_dataset (train.py:150-176).  `device`: keep the windowed dataset resident in that GPU's HBM
    (batches are device-side gathers; what `main` does) instead of yielding numpy batches.  --use_tfm: time and frequency masks (transforms.mask, per
    sample and per 100-frame segment); --use_acs: foa_intensity_vec_aug on the batch — both on the device
    (seld_amd.transforms), applied by `iterloop` to the batch after its host->HBM copy."""
    import os
    from . import data_loader as dl, transforms as tfm
    path = os.path.join(config.abspath, 'DCASE2021/feat_label/')
    x, y = dl.load_seldnet_data(os.path.join(path, 'foa_dev_norm'), os.path.join(path, 'foa_dev_label'), mode=mode, n_freq_bins=64)
    device_transforms = []
    if getattr(config, 'use_tfm', False) and mode == 'train':
        device_transforms.append(lambda x, y, rng: (tfm.mask(x, -3, max_mask_size=config.time_mask_size, rng=rng), y))
        device_transforms.append(lambda x, y, rng: (tfm.mask(x, -2, max_mask_size=config.freq_mask_size, rng=rng), y))
    if getattr(config, 'use_acs', False) and mode == 'train':
        device_transforms.append(lambda x, y, rng: tfm.foa_intensity_vec_aug(x, y, rng=rng))
    ds = dl.seldnet_data_to_dataloader(x, y, train=mode == 'train', label_window_size=60, batch_size=config.batch,
                                       loop_time=config.loop_time, device_transforms=device_transforms)
    return ds.to_device(device) if device is not None else ds


def iterloop(model: SeldNet, dataset, sed_loss, doa_loss, metric_class, config, optimizer=None, mode='train',
             process_group=None):
    """The step loop of reference train.iterloop (train.py:47-147) -> (mean sed loss, mean doa loss, seld score).
    `metric_class` (seld_amd.metrics.SELDMetrics or None) is updated on the device after every step, as the
    reference does on the host (train.py:82-83); csv dumps / DCASE official metrics / tensorboard
    (train.py:85-145) are outside the accelerated path.  One host sync per epoch instead of one per step."""
    from . import metrics as _metrics
    loss_weight = [int(i) for i in config.loss_weight.split(',')]
    tot_s = torch.zeros((), device=model._dev)
    tot_d = torch.zeros((), device=model._dev)
    n = 0
    dev_tf = getattr(dataset, 'device_transforms', None)
    for x, y in dataset:
        if dev_tf:        # augmentation on the device batch; labels arrive unsplit [b,60,4C] (data_loader.SeldDataset)
            x = torch.as_tensor(x, dtype=torch.float32).to(model._dev).contiguous()
            y = torch.as_tensor(y, dtype=torch.float32).to(model._dev).contiguous()
            for f in dev_tf:
                x, y = f(x, y, dataset.rng)
            nc = y.shape[-1] // 4
            y = (y[..., :nc].contiguous(), y[..., nc:].contiguous())
        if mode == 'train':
            preds, sloss, dloss = trainstep(model, x, y, sed_loss, doa_loss, loss_weight, optimizer, config.agc, process_group)
        else:
            preds, sloss, dloss = teststep(model, x, y, sed_loss, doa_loss)
        if metric_class is not None:
            metric_class.update_states(y, preds)
        tot_s += sloss
        tot_d += dloss.mean()
        n += 1
    score = _metrics.calculate_seld_score(metric_class.result()) if metric_class is not None else float('nan')
    return float(tot_s.item()) / max(n, 1), float(tot_d.item()) / max(n, 1), score


def main(config, model_config=None, max_epochs=None):
    """reference train.main (train.py:264-390) around the accelerated steps: datasets (train / val / test folds), model with
    n_classes forced to 12, `--resume` from the saved weights, Adam, BCE + MSE|MMSE, SELDMetrics, and the epoch loop —
    train, validation and evaluation passes — with best-model save, LR decay on plateau and early stopping on the
    validation SELD score (block-wise metrics.SELDMetrics score; the reference's csv-based DCASE scorer is out of scope)."""
    import os
    from . import metrics as _metrics
    from . import models
    if isinstance(config, tuple):
        config, model_config = config
    dev = torch.device('cuda', torch.cuda.current_device())
    trainset, valset = get_dataset(config, 'train', dev), get_dataset(config, 'val', dev)    # HBM-resident
    testset = get_dataset(config, 'test', dev)                                                # train.py:290 (fold 6)
    x, y = next(iter(trainset.take(1)))
    input_shape = (max(config.batch, valset.batch_size, testset.batch_size),) + tuple(x.shape[1:])
    model_config = dict(model_config)
    n_classes = 12                                                  # train.py:306-307
    model_config['n_classes'] = n_classes
    model = getattr(models, config.model)(input_shape, model_config)
    model.summary()
    optimizer = Adam(config.lr)
    sed_loss = losses.BinaryCrossentropy()
    doa_loss = losses.get_doa_loss(config.doa_loss)
    metric_class = _metrics.SELDMetrics(doa_threshold=config.lad_doa_thresh, n_classes=n_classes)
    model_path = os.path.join('./saved_model', config.name)
    os.makedirs(model_path, exist_ok=True)
    if getattr(config, 'resume', False):                           # train.py:322-331: the saved model's weights, not the optimizer's slots
        from glob import glob
        saved = sorted(glob(os.path.join(model_path, '*.npz')))
        if len(saved) == 0:
            raise ValueError('the model is not existing, resume fail')
        model.load_weights(saved[0])
    best, early, lr_pat, history = 99999.0, 0, 0, []
    for epoch in range(config.epoch if max_epochs is None else min(config.epoch, max_epochs)):
        metric_class.reset_states()
        tr = iterloop(model, trainset, sed_loss, doa_loss, metric_class, config, optimizer, 'train')
        metric_class.reset_states()
        va = iterloop(model, valset, sed_loss, doa_loss, metric_class, config, mode='val')
        score = va[2]
        metric_class.reset_states()                                 # evaluation loop (train.py:367-369)
        te = iterloop(model, testset, sed_loss, doa_loss, metric_class, config, mode='test')
        history.append({'epoch': epoch, 'train': tr, 'val': va, 'test': te, 'score': score, 'lr': optimizer.learning_rate})
        print(f'epoch {epoch}: train sed/doa/seld {tr[0]:.4f}/{tr[1]:.5f}/{tr[2]:.4f}  val {va[0]:.4f}/{va[1]:.5f}/{va[2]:.4f}'
              f'  test {te[0]:.4f}/{te[1]:.5f}/{te[2]:.4f}')
        if best > score:                                            # train.py:372-380
            old = os.path.join(model_path, f'bestscore_{best}.npz')
            if os.path.exists(old):
                os.remove(old)
            best, early, lr_pat = score, 0, 0
            model.save_weights(os.path.join(model_path, f'bestscore_{best}.npz'))
        else:                                                       # train.py:381-390
            if lr_pat == config.lr_patience and config.decay != 1:
                optimizer.learning_rate *= config.decay
                lr_pat = 0
            if early == config.patience:
                print(f'Early Stopping at {epoch}, score is {score}')
                break
            early += 1
            lr_pat += 1
    return model, history
