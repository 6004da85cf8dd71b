"""Shared by tests/test_cpu_driver_contract.py and tests/test_gpu_driver_contract.py: stand-ins for the reference's CPU input pipeline
(``get_dataset`` / ``get_data_loader`` of semilearn/core/utils/build.py:60,:121 -- same signatures, same return shapes) and ``main_worker``'s
call sequence on the algorithm object (/root/reference/train.py:394-431), written once so that both tests replay the SAME lines."""
import logging

import numpy as np
import torch

CALLS = []


class _Dset:
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)


def get_dataset(args, algorithm, dataset, num_labels, num_classes, data_dir="./data", include_lb_to_ulb=True):
    """Stand-in: [n_lb] labelled and [n_ulb] unlabelled synthetic images (weak / strong views as two noise draws), an eval split, no test split."""
    CALLS.append(("get_dataset", algorithm, dataset, num_labels, num_classes, data_dir, include_lb_to_ulb))
    rng = np.random.Generator(np.random.PCG64(5))
    S = args.img_size
    mk = lambda n: rng.standard_normal((n, 3, S, S)).astype(np.float32)     # noqa: E731
    n_lb, n_ulb = num_labels, args.stand_in_ulb
    lb = [dict(idx_lb=i, x_lb=x, y_lb=int(rng.integers(num_classes))) for i, x in enumerate(mk(n_lb))]
    ulb = [dict(idx_ulb=i, x_ulb_w=w, x_ulb_s=s) for i, (w, s) in enumerate(zip(mk(n_ulb), mk(n_ulb)))]
    ev = [dict(x_lb=x, y_lb=int(rng.integers(num_classes))) for x in mk(6)]
    return {"train_lb": _Dset(lb), "train_ulb": _Dset(ulb), "eval": _Dset(ev), "test": None}


def get_data_loader(args, dset, batch_size=None, shuffle=False, num_workers=4, pin_memory=False, data_sampler="RandomSampler", num_epochs=None,
                    num_iters=None, generator=None, drop_last=True, distributed=False):
    """Stand-in: a re-iterable of collated dict batches in dataset order (``drop_last`` honoured; train loaders wrap around to ``num_iters``)."""
    CALLS.append(("get_data_loader", len(dset), batch_size, data_sampler, num_iters, num_epochs, num_workers, drop_last, distributed))

    class Loader:
        def __iter__(self_):
            n = len(dset)
            total = num_iters if num_iters is not None else (n // batch_size if drop_last else -(-n // batch_size))
            for it in range(total):
                rows = [dset.items[(it * batch_size + j) % n] for j in range(batch_size)] if num_iters is not None else \
                    dset.items[it * batch_size:(it + 1) * batch_size]
                yield {k: torch.as_tensor(np.stack([r[k] for r in rows])) for k in rows[0]}

        def __len__(self_):
            return num_iters if num_iters is not None else len(dset) // batch_size
    return Loader()


def main_worker_tail(args, get_algorithm, net_builder, count_parameters, send_model_cuda, load_path=None):
    """train.py:389-431 from ``_net_builder = get_net_builder(...)`` on, with the two engine-side replacements INTEGRATION.md names
    (count_parameters needs none: the reference's own expression is passed in as well and must agree)."""
    logger = logging.getLogger("driver-replay")
    model = get_algorithm(args, net_builder, None, logger)                                      # :394
    n_params = count_parameters(model.model)                                                    # :396
    logger.info(f"Number of Trainable Params: {n_params}")
    model.model = send_model_cuda(args, model.model)                                            # :399
    model.ema_model = send_model_cuda(args, model.ema_model, clip_batch=False)                  # :400
    if getattr(args, "resume", False) and load_path is not None:                                # :404-409
        model.load_model(load_path)
    if hasattr(model, "warmup"):                                                                # :413
        model.warmup()
    model.train()                                                                               # :419
    results = getattr(model, "results_dict", {})                                                # :422
    if hasattr(model, "finetune"):                                                              # :425
        model.finetune()
    return model, n_params, results
