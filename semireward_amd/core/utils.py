"""The two helpers the reference's driver calls on ``model.model`` between ``get_algorithm`` and ``model.train()`` (train.py:396-400), for engine
models.  ``count_parameters`` is the reference's own expression and works unchanged through ``ModuleSurface.parameters()``; ``send_model_cuda`` has
to be THIS one: the reference's ends in ``DistributedDataParallel(model)`` (semilearn/core/utils/misc.py:56-63), which needs an autograd
``nn.Module`` whose gradients appear through backward hooks -- the engine's backward is hand-written and its gradient exchange is one flat block
(semireward_amd/distributed.py, ParamUpdateHook).  INTEGRATION.md shows the edit of train.py."""
import torch


def count_parameters(model):
    """semilearn/core/utils/misc.py:73-75."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def send_model_cuda(args, model, clip_batch=True):
    """semilearn/core/utils/misc.py:39-70 for an engine model: same argument meaning, same side effects on ``args`` (the per-node batch size
    becomes per-GPU under ``distributed``), same errors -- and no wrapper object: data parallel is the engine's own (every rank all-reduces its
    flat gradient block in ParamUpdateHook; a BatchNorm backbone exchanges its statistics itself = SyncBatchNorm, nets/wrn.py), so the returned
    model is the model, and ``model.model.module`` of the reference's DDP case does not exist."""
    if not torch.cuda.is_available():
        raise Exception("ONLY GPU TRAINING IS SUPPORTED")
    if getattr(args, "distributed", False):
        ngpus_per_node = torch.cuda.device_count()
        if args.gpu is not None:
            torch.cuda.set_device(args.gpu)
            if clip_batch:
                args.batch_size = int(args.batch_size / ngpus_per_node)      # batch_size per node -> per GPU (misc.py:48-53)
            model.cuda(args.gpu)
        else:
            model.cuda()
    elif getattr(args, "gpu", None) is not None:
        torch.cuda.set_device(args.gpu)
        model = model.cuda(args.gpu)
    else:
        # misc.py:67-68 wraps in torch.nn.DataParallel (ONE process scattering batches over all visible GPUs): not a mode of the engine, which
        # is one process per GPU like the reference's own multi-GPU path (train.py:344)
        raise NotImplementedError("single-process multi-GPU (torch.nn.DataParallel, args.gpu None without args.distributed) is not supported by the "
                                  "HIP engine: pass --gpu <id>, or start one process per GPU (--multiprocessing-distributed)")
    return model


def reference_data_functions():
    """(get_dataset, get_data_loader) of the reference when ``semilearn`` is importable (the maintainer's tree: semilearn/core/utils/build.py:60,
    :121), else (None, None): the CPU input pipeline (datasets, samplers, collators) is the reference's, not rebuilt here."""
    try:
        from semilearn.core.utils import get_data_loader, get_dataset
    except Exception:       # noqa: BLE001  (absent, or its own imports -- torchvision, ruamel -- are)
        return None, None
    return get_dataset, get_data_loader
