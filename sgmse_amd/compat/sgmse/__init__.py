"""Import alias so that code written against the reference package name (`from sgmse.model import ScoreModel`,
`from sgmse.util.other import pad_spec`, pickled `sgmse.data_module.SpecsDataModule` inside Lightning checkpoints)
resolves to sgmse_amd.  Put `sgmse_amd/compat` on PYTHONPATH ahead of (or instead of) the reference tree."""
import sys

import sgmse_amd
from sgmse_amd import backbones, data_module, model, sampling, sdes, util
from sgmse_amd.backbones import ncsnpp as _ncsnpp, shared as _shared
from sgmse_amd.sampling import correctors as _correctors, predictors as _predictors
from sgmse_amd.util import other as _other, registry as _registry

for _name, _mod in {
    "model": model, "sdes": sdes, "sampling": sampling, "backbones": backbones, "data_module": data_module, "util": util,
    "util.other": _other, "util.registry": _registry, "backbones.ncsnpp": _ncsnpp, "backbones.shared": _shared,
    "sampling.predictors": _predictors, "sampling.correctors": _correctors,
}.items():
    sys.modules[f"{__name__}.{_name}"] = _mod
    if "." not in _name:
        globals()[_name] = _mod
__version__ = sgmse_amd.__version__
