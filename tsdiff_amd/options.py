"""The switches of the host side, in ONE documented object.

Every switch selects between forms that compute the same function (bit-identical unless noted); they exist for
cross-check tests and A/B measurements, a user of the package never needs them.  The defaults can be overridden once,
at import, by `TSDIFF_*` environment variables (read here and nowhere else); tests and tools flip the attributes of
`tsdiff_amd.options.OPTIONS` (e.g. `monkeypatch.setattr(OPTIONS, "one_launch", False)`), which every later call sees.

| attribute          | env variable              | default | meaning |
|--------------------|---------------------------|---------|---------|
| gemm               | TSDIFF_GEMM               | "h2"    | arithmetic of the inference forward's tile GEMMs: "h2" = split-f16 operands on the f16 MFMA pipes (22-bit operands, fp32 accumulation; csrc/split16.hpp), "f32" = fp32-input MFMA.  NOT bit-identical (1e-6 of the tensor scale apart).  A call that leaves the f16 range reruns in "f32" by itself |
| typed_tiles        | TSDIFF_TYPED_TILES        | True    | edge embedding on static type-sorted tiles with per-type folded matrices (kernels_typed.hip); False: the generic embedding kernel (and, since the split-f16 forward needs the typed tiles, fp32 arithmetic) |
| one_launch         | TSDIFF_ONE_LAUNCH         | True    | small batches: the L interaction blocks + pair MLP as ONE launch (forward_mega_kernel); False: one launch per block |
| wide_filter_tiles  | TSDIFF_WIDE_FILTER_TILES  | True    | 64-row filter tiles where a block launch is many chip-fulls deep, 64-row filter and pair tiles in the one-launch kernel where they fill the chip, 64-row tiles in a stand-alone pair launch of >= 4096 tiles; False: 32-row tiles everywhere.  Bit-identical |
| fused_step_tail    | TSDIFF_FUSED_TAIL         | True    | sampling loop: update + next step's edge lists as one launch; False: three launches |
| fused_encoder      | TSDIFF_FUSED_ENCODER      | True    | chip-full launches (big batches, ensembles): the whole SchNet encoder as ONE launch of per-unit workgroups that never write the CFConv filters to memory (kernels_unit.hip); False: one launch per block with materialised filters; "force": also where the one-launch form would apply.  Bit-identical (the messages are added in the directed list's order) |
| train              | TSDIFF_TRAIN              | "fused" | training step: "fused" = forward + loss and the whole backward as two library calls (csrc/train_step.hip), "ops" = one autograd node per operation (same kernels; the cross-check) |
| train_gemm         | TSDIFF_TRAIN_GEMM         | "h2"    | arithmetic of the fused training step's tile GEMMs: "f32" = fp32-input MFMA, "h2" = split-f16 operands (gradient operands scaled by a power of two per tensor).  A step whose activations leave the f16 range is recomputed in "f32" |
| train_fallback_latch | TSDIFF_TRAIN_FALLBACK_LATCH | 16 | split-f16 training step: a step whose activations leave the f16 range is recomputed in fp32 and the NEXT step tries split-f16 again (`model._h2_range_trips` counts the trips); this many trips IN A ROW latch the model to fp32 (each trip costs a second forward) |
| train_flat_grad    | TSDIFF_TRAIN_FLAT_GRAD    | True    | fused training step with parameters that are views of one flat buffer (optim.get_optimizer) and carry no .grad: autograd sees ONE flat leaf instead of ~80 parameters, and the backward hands out cached views of a persistent flat gradient as the Parameters' .grad (−0.5 ms of host time per step).  Same gradients, bit for bit; the .grad tensors of consecutive steps share memory (as with zero_grad(set_to_none=False)) and torch.autograd.grad(loss, parameters) does not see the parameters; False: one autograd input per parameter |
| dp_overlap         | TSDIFF_DP_OVERLAP         | False   | data-parallel training step (distributed.dp_backward): all-reduce the interaction blocks' gradient range (83 % of the flat vector) early on a side stream beside the rest of the backward pass, head and tail behind it (three collectives); False: ONE all-reduce of the flat gradient behind the backward pass.  Off by default since round 6: +0.14 ms of host time per step against <= 0.11 ms of hidden communication on 8 ranks over xGMI while the step is host-bound.  Same gradients either way |
| train_side_lane    | TSDIFF_TRAIN_SIDE_LANE    | True    | split-f16 training step: the backward's small latency-bound gradient launches (embedding tables, narrow layers) run on a stream of the library's own beside the batched weight gradients; False: everything on the caller's stream.  Bit-identical |
"""
import os
from dataclasses import dataclass


def _flag(name, default=True):
    v = os.environ.get(name)
    return default if v is None else v != "0"


@dataclass
class Options:
    gemm: str = "h2"
    typed_tiles: bool = True
    one_launch: bool = True
    wide_filter_tiles: bool = True
    fused_step_tail: bool = True
    fused_encoder: bool = True
    train: str = "fused"
    train_gemm: str = "h2"
    train_fallback_latch: int = 16
    train_side_lane: bool = True
    dp_overlap: bool = False
    train_flat_grad: bool = True

    @classmethod
    def from_env(cls):
        o = cls(gemm=os.environ.get("TSDIFF_GEMM", "h2"), typed_tiles=_flag("TSDIFF_TYPED_TILES"),
                one_launch=_flag("TSDIFF_ONE_LAUNCH"), wide_filter_tiles=_flag("TSDIFF_WIDE_FILTER_TILES"),
                fused_step_tail=_flag("TSDIFF_FUSED_TAIL"), fused_encoder=_flag("TSDIFF_FUSED_ENCODER"),
                train=os.environ.get("TSDIFF_TRAIN", "fused"), train_gemm=os.environ.get("TSDIFF_TRAIN_GEMM", "h2"),
                train_fallback_latch=int(os.environ.get("TSDIFF_TRAIN_FALLBACK_LATCH", "16")),
                train_side_lane=_flag("TSDIFF_TRAIN_SIDE_LANE"), dp_overlap=_flag("TSDIFF_DP_OVERLAP", False),
                train_flat_grad=_flag("TSDIFF_TRAIN_FLAT_GRAD"))
        o.validate()
        return o

    def validate(self):
        if self.gemm not in ("h2", "f32"):
            raise ValueError(f"TSDIFF_GEMM={self.gemm!r}: expected 'h2' or 'f32'")
        if self.train_gemm not in ("h2", "f32"):
            raise ValueError(f"TSDIFF_TRAIN_GEMM={self.train_gemm!r}: expected 'h2' or 'f32'")
        if self.train_fallback_latch < 1:
            raise ValueError(f"TSDIFF_TRAIN_FALLBACK_LATCH={self.train_fallback_latch!r}: expected an integer >= 1")
        if self.train not in ("fused", "ops"):
            raise ValueError(f"TSDIFF_TRAIN={self.train!r}: expected 'fused' or 'ops'")


OPTIONS = Options.from_env()
