"""MI355X-native NGCF embedding-propagation engine.

Drop-in for the hot path of haesungpyun/seoul_tourism_recommendation_NGCF:
`from seoul_tourism_recommendation_ngcf_amd import NGCF, BPR` replaces
`from NGCF import NGCF` / `from bprloss import BPR` in main.py, experiment.py and demo.py.
"""
from .NGCF import NGCF
from .bprloss import BPR
from .graphed import GraphedForward, GraphedTrainStep
from . import engine, graphs

__all__ = ["NGCF", "BPR", "GraphedForward", "GraphedTrainStep", "engine", "graphs"]
