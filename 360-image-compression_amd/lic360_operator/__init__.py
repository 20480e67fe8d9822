"""`lic360_operator` -- the reference's operator package surface (lic360_operator/__init__.py:1-29)
for the hot path, on top of the HIP-backed `lic360` module.  Training / metric utilities of the
reference (GDN, SSIM, MultiProject, MaskConv2, DropGrad, Logger, ModuleSaver) are out of scope
 and not provided."""
from .base import BaseOpModule
from .quantize import ImpMap, QUANT, Dquant, Dtow, Imp2mask, Scale, ContextReshape, ContextShift
from .tables import EntropyGmm, EntropyGmmTable, EntropyBatchGmmTable, EntropyTable
from .sphere import SpherePad, SphereTrim, SphereCutEdge, SphereLatScaleNet
from .planes import CodeContex, TileExtract, TileExtractBatch, TileInput, TileAdd
from .conv import CconvDc, CconvDcBatch, CconvEc, CconvEcBatch
