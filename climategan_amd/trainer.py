"""Host-side mirror of the inference half of the reference's ``climategan/trainer.py``:
``Trainer.setup(inference=True)``, ``infer_all`` and ``compute_flood`` -- the stage order, binarisation and uint8
conversion of the reference, every arithmetic step a HIP kernel behind the C ABI.

Built: the flood event (Masker -> mask -> Painter).  Smog and wildfire (rows N1, ``trainer.py:1821-1842,1879-1939``)
are not built yet: asking for them raises NotImplementedError instead of silently skipping.
"""
import time

import torch

from . import ops
from .generator import create_generator
from .utils import find_target_size


class Timer:
    """reference utils.py:899-960: context manager appending elapsed seconds to ``store`` (device-synchronised)."""

    def __init__(self, name="", store=None, precision=3, ignore=False, cuda=True):
        self.store = store
        self.cuda = cuda and torch.cuda.is_available()
        self.ignore = ignore

    def __enter__(self):
        if not self.ignore:
            if self.cuda:
                torch.cuda.synchronize()
            self._t = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if not self.ignore:
            if self.cuda:
                torch.cuda.synchronize()
            if self.store is not None:
                self.store.append(time.perf_counter() - self._t)
        return False


class Trainer:
    """Inference-side subset of the reference Trainer (trainer.py:63-216): owns ``G``; no logger / comet / data."""

    def __init__(self, opts, comet_exp=None, verbose=0, device=None):
        self.opts = opts
        self.verbose = verbose
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda:0" if torch.cuda.is_available() else "cpu")
        self.G = None
        self.D = None
        self.is_setup = False
        self.has_painter = "p" in opts.tasks

    def setup(self, inference=False):
        """reference trainer.py:701-760 (inference branch: generator only)."""
        if not inference:
            raise NotImplementedError("Trainer.setup(inference=False): the training harness (SURVEY row H2) is not built")
        self.G = create_generator(self.opts, device=self.device, no_init=True, verbose=self.verbose)
        if self.has_painter:
            self.G.painter.set_latent_shape(find_target_size(self.opts, "x"), True)       # trainer.py:727-728
        self.G.eval()
        self.is_setup = True
        return self

    # ------------------------------------------------------------------------------------------ events
    def compute_flood(self, x, z=None, z_depth=None, m=None, s=None, cloudy=None, bin_value=-1):
        """reference trainer.py:1844-1877"""
        if m is None:
            if z is None:
                z = self.G.encode(x)
            if "d" in self.opts.tasks and self.opts.gen.m.use_dada and z_depth is None:
                _, z_depth = self.G.decoders["d"].forward_nhwc(z)
            m = self.G.mask(x=x, z=z, z_depth=z_depth)
        if bin_value >= 0:
            m = ops.binarize(m, bin_value)                                                # (m > bin_value).to(m.dtype)
        if cloudy:
            assert s is not None
            return self.G.paint_cloudy(m, x, s)
        return self.G.paint(m, x)

    def compute_fire(self, x, seg_preds=None, z=None, z_depth=None):
        raise NotImplementedError("wildfire event (trainer.py:1821-1842, fire.py) has no HIP path yet (SURVEY row N1)")

    def compute_smog(self, x, z=None, d=None, s=None, use_sky_seg=False):
        raise NotImplementedError("smog event (trainer.py:1879-1939) has no HIP path yet (SURVEY row N1)")

    @torch.no_grad()
    def infer_all(self, x, numpy=True, stores={}, bin_value=-1, half=False, xla=False, cloudy=False,
                  auto_resize_640=False, ignore_event=set(), return_masks=False):
        """reference trainer.py:217-334.  ``half`` selects fp16 I/O tensors (the kernels compute in 16-bit either
        way); ``xla`` is accepted and ignored.  Events not in ``ignore_event`` must have a HIP path."""
        assert self.is_setup
        assert len(x.shape) in {3, 4}, f"Unknown Data shape {x.shape}"
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(x, device=self.device)
        if len(x.shape) == 3:
            x = x.unsqueeze(0)
        if x.shape[1] != 3:
            assert x.shape[-1] == 3, f"Unknown x shape to permute {x.shape}"
            x = x.permute(0, 3, 1, 2)
        if x.device != self.device:
            x = x.to(self.device)
        if auto_resize_640 and (x.shape[-1] != 640 or x.shape[-2] != 640):
            raise NotImplementedError("auto_resize_640: the input-side resize (SURVEY row N3) has no HIP path yet")
        x = x.half() if half else x.float()
        x = x.contiguous()

        self.G.painter.set_latent_shape(x.shape, True)                                   # trainer.py:266

        with Timer(store=stores.get("all events", [])):
            with Timer(store=stores.get("encode", [])):
                z = self.G.encode(x)
            with Timer(store=stores.get("depth", [])):
                depth_nhwc, z_depth = self.G.decoders["d"].forward_nhwc(z)
            with Timer(store=stores.get("segmentation", [])):
                seg_nhwc = self.G.decoders["s"].forward_nhwc(z, z_depth)
            with Timer(store=stores.get("mask", [])):
                # make_m_cond (trainer.py:285) only matters for the SPADE mask decoder, which has no HIP path yet
                mask = self.G.mask(z=z, cond=None, z_depth=z_depth).to(x.dtype)

            wildfire = smog = flood = None
            if "wildfire" not in ignore_event:
                with Timer(store=stores.get("wildfire", [])):
                    wildfire = self.compute_fire(x, seg_preds=seg_nhwc)
            if "smog" not in ignore_event:
                with Timer(store=stores.get("smog", [])):
                    smog = self.compute_smog(x, d=depth_nhwc, s=seg_nhwc)
            if "flood" not in ignore_event:
                with Timer(store=stores.get("flood", [])):
                    flood = self.compute_flood(x, m=mask, s=seg_nhwc, cloudy=cloudy, bin_value=bin_value)

        output_data = {}
        with Timer(store=stores.get("numpy", []), ignore=not numpy):
            for name, ev in (("flood", flood), ("wildfire", wildfire), ("smog", smog)):
                if ev is None:
                    continue
                if numpy:
                    ev = ops.normalize_to_uint8(ev).cpu().numpy()                        # trainer.py:311-326
                output_data[name] = ev
        if return_masks:
            output_data["mask"] = ops.binarize(mask, bin_value, want_float=False, want_uint8=True).cpu().numpy()
        return output_data
