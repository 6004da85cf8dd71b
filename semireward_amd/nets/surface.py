"""What the reference's DRIVER touches on ``model.model`` before training starts, for the engine backbones (flat parameter block + launches, no
nn.Module): train.py:396 ``count_parameters(model.model)`` (semilearn/core/utils/misc.py:73-75 iterates ``model.parameters()`` and reads
``requires_grad`` / ``numel()``) and train.py:399-400 ``send_model_cuda`` (misc.py:39-70: ``model.cuda(gpu)``,
``nn.SyncBatchNorm.convert_sync_batchnorm(model)`` which walks ``named_children()``, then the DistributedDataParallel wrap -- the one step an engine
model cannot take and ``semireward_amd.core.utils.send_model_cuda`` replaces; INTEGRATION.md names the two lines of train.py)."""
import torch


class ModuleSurface:
    frozen_params = ()          # names whose gradient is None in the reference (no path to the loss): requires_grad False here

    def parameters(self, recurse=True):
        """Leaf views of the flat parameter block, one per reference parameter, in ``named_parameters()`` order: ``requires_grad`` as in the
        reference module, ``.grad`` = the matching view of the flat gradient block (so ``p.grad`` reads what the hand-written backward
        accumulated).  Fresh view objects per call -- the engine itself never touches them (its launches take raw pointers)."""
        for n, s in self.names_shapes:
            o = self.offsets[n][0]
            k = int(torch.Size(s).numel())
            p = self.flat[o:o + k].view(s)
            if n not in self.frozen_params:
                p.requires_grad_(True)
                p.grad = self.grad[o:o + k].view(s)
            yield p

    def _same_device(self, device):
        if device is None:
            return
        d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if d.type != self.device.type or (d.index is not None and self.device.index is not None and d.index != self.device.index):
            raise RuntimeError("%s lives on %s (its blocks are allocated where it is built: pass device= to the builder); it cannot be moved to %s"
                               % (type(self).__name__, self.device, d))

    def cuda(self, device=None):
        """nn.Module.cuda(gpu) of misc.py:50/59/65: the engine model is built on its GPU already -- checked, not moved."""
        self._same_device(device if device is not None else "cuda")
        return self

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", next((a for a in args if isinstance(a, (str, int, torch.device))), None))
        if any(isinstance(a, torch.dtype) for a in args) or kwargs.get("dtype") is not None:
            raise RuntimeError("the engine has one numeric mode (fp32 master block, bf16 operands): .to(dtype) is not supported")
        self._same_device(dev)
        return self

    # nn.SyncBatchNorm.convert_sync_batchnorm(model) (misc.py:55) walks named_children() and returns the module itself when there is no
    # nn.BatchNorm child: the engine's BatchNorm backbone exchanges its statistics itself under data parallel (nets/wrn.py ``dp``)
    def named_children(self):
        return iter(())

    def children(self):
        return iter(())

    def modules(self):
        return iter((self,))
