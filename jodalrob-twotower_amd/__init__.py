"""MI355X-native two-tower training step behind the reference's Python API
(zoongahn/jodalroB-twoTower: TwoTowerTrainTask / TwoTowerModel / towers / CategoricalEmbedder /
FeaturePreprocessor).  Host code is Python on PyTorch-ROCm tensors; every hot-path operation goes
through the C ABI of libtwotower_hip.so (include/twotower.h) into hand-written gfx950 HIP kernels.

The directory name carries a hyphen; import it as `jodalrob_twotower_amd` (the alias module at the
repository root registers it).
"""
from .kjt import KeyedJaggedTensor, build_batch_kjt                               # noqa: F401
from .schema import (PairSchema, SideSchema, TorchRecSchema, build_torchrec_schema_from_meta,   # noqa: F401
                     classify_columns)
from .cat_embed import CategoricalEmbedder, EmbeddingStore, create_categorical_embedder         # noqa: F401
from .towers import BaseTower, CompanyTower, NoticeTower                          # noqa: F401
from .two_tower_model import TwoTowerModel, create_two_tower_model               # noqa: F401
from .two_tower_train_task import TwoTowerTrainTask, create_two_tower_train_task  # noqa: F401
from .feature_projector import FeatureProjector                                   # noqa: F401
from .feature_preprocessor import FeaturePreprocessor                             # noqa: F401
from .evaluator import TwoTowerEvaluator                                          # noqa: F401
from .optim import FusedAdam                                                      # noqa: F401

__all__ = ["KeyedJaggedTensor", "build_batch_kjt", "SideSchema", "PairSchema", "TorchRecSchema",
           "build_torchrec_schema_from_meta", "classify_columns", "CategoricalEmbedder", "EmbeddingStore",
           "create_categorical_embedder", "BaseTower", "NoticeTower", "CompanyTower", "TwoTowerModel",
           "create_two_tower_model", "TwoTowerTrainTask", "create_two_tower_train_task", "FeatureProjector",
           "FeaturePreprocessor", "TwoTowerEvaluator", "FusedAdam"]
