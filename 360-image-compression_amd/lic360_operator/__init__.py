"""`lic360_operator` -- the reference's operator package surface (lic360_operator/__init__.py:1-29), every name of it, on
top of the HIP-backed `lic360` module.  The hot-path operators go through the C ABI; the utilities in extras.py are plain
torch (GDN, DropGrad, SSIM, ModuleSaver, Logger) or thin modules over native ops (MultiProject over ProjectsOp, MaskConv2 =
torch conv2d over a weight masked by MaskConstrainOp)."""
from .base import BaseOpModule
from .quantize import ImpMap, QUANT, Dquant, Dtow, Imp2mask, Scale, ContextReshape, ContextShift
from .tables import EntropyGmm, EntropyGmmTable, EntropyBatchGmmTable, EntropyTable
from .sphere import SpherePad, SphereTrim, SphereCutEdge, SphereLatScaleNet
from .planes import CodeContex, TileExtract, TileExtractBatch, TileInput, TileAdd
from .conv import CconvDc, CconvDcBatch, CconvEc, CconvEcBatch
from .extras import GDN, DropGrad, SSIM, ModuleSaver, Logger, MultiProject, MaskConv2
