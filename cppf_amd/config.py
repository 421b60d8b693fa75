"""Path parameters of the reference's hydra configs (config/config.yaml, config/category/*.yaml):
only the keys the hot path reads."""
from dataclasses import dataclass, field
from typing import List


@dataclass
class CategoryConfig:
    category: str
    res: float
    vote_range: List[float]
    scale_mean: List[float]
    regress_right: bool
    up_sym: bool = False
    z_right: bool = False
    tr_num_bins: int = 32          # config/config.yaml:7
    rot_num_bins: int = 36         # config/config.yaml:8
    knn: int = 60                  # config/config.yaml:22
    ppffcs: List[int] = field(default_factory=lambda: [84, 32, 32, 16])   # train.py:35

    @property
    def out_dim(self):             # train.py:35
        return 2 * self.tr_num_bins + 2 * self.rot_num_bins + 2 + 3


def _c(cat, res, vr, sm, rr, us=False):
    return CategoryConfig(cat, res, [vr, vr], sm, rr, us)


# config/category/{bottle,bowl,camera,can,laptop,mug}.yaml (NOCS) and the SUN RGB-D ones
CATEGORIES = {
    "bottle": _c("bottle", 4e-3, 0.25, [0.05, 0.15, 0.05], False, True),
    "bowl": _c("bowl", 4e-3, 0.12, [0.07, 0.03, 0.07], False),
    "camera": _c("camera", 4e-3, 0.15, [0.05, 0.05, 0.07], True),
    "can": _c("can", 4e-3, 0.1, [0.037, 0.055, 0.037], False, True),
    "laptop": _c("laptop", 1e-2, 0.3, [0.13, 0.1, 0.15], True),
    "mug": _c("mug", 4e-3, 0.12, [0.06, 0.05, 0.045], True, True),
    "bathtub": _c("bathtub", 3e-2, 1.104495769458527, [0.3886107936507936, 0.23178569841269842, 0.6773600634920637], True),
    "bed": _c("bed", 3e-2, 1.860647331329598, [1.0274129618644066, 0.49417050423728714, 0.7768076129943503], True),
    "bookshelf": _c("bookshelf", 3e-2, 1.501261827212904, [0.2166509734042553, 0.8096084361702133, 0.6821934095744682], True, True),
    "chair": _c("chair", 3e-2, 0.7863312261283193, [0.29636475108453486, 0.4208450279047945, 0.2789450356430997], True),
    "sofa": _c("sofa", 3e-2, 1.5296381674297101, [0.48090147859327237, 0.4228677186544341, 0.9288330076452599], True),
    "table": _c("table", 3e-2, 1.2363029403529906, [0.4305247031772566, 0.35199163168896463, 0.6905431275083608], True),
}
NOCS_CATEGORIES = ["bottle", "bowl", "camera", "can", "laptop", "mug"]   # nocs/inference.py synset order
