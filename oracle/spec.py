"""Sizes of the Q-network (ORACLE, test infrastructure only).

Follows `/root/reference/BS_brain.py:94-104` (size bookkeeping of `BS.__init__`) and the
generalisation table in SURVEY.md Appendix E.
"""
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class GnnSpec:
    n_nodes: int = 4          # N  = num_D2D                     (BS_brain.py:95)
    n_channels: int = 4       # C  = num_CH                      (BS_brain.py:97)
    feat_dim: int = 16        # F  = num_Feedback                (BS_brain.py:98)
    n_mp_layers: int = 2      # L  message-passing stages after the embed (BS_brain.py:154-164)
    hidden: Tuple[int, int, int] = (80, 40, 20)   # Dense widths (BS_brain.py:176-178)
    share_weights: bool = False  # reference: one weight set PER NODE (BS_brain.py:121-200)
    input_node_info: int = 3  # BS_brain.py:294
    input_edge_info: int = 1  # BS_brain.py:295
    n_neighbor: int = 1       # BS_brain.py:96

    @property
    def node_in(self) -> int:   # Dn = num_One_Node_Input  (BS_brain.py:101)
        return ((self.input_node_info - 1) * self.n_channels + 1) * self.n_neighbor

    @property
    def edge_in(self) -> int:   # De = num_One_Edge_Input  (BS_brain.py:102)
        return self.input_edge_info * self.n_channels

    @property
    def n_slots(self) -> int:   # number of independent weight sets
        return 1 if self.share_weights else self.n_nodes

    @property
    def dense_dims(self):       # (in, out) of the 4 Dense layers (BS_brain.py:175-179)
        F, Dn, C = self.feat_dim, self.node_in, self.n_channels
        h1, h2, h3 = self.hidden
        return [(Dn + 2 * F, h1), (h1, h2), (h2, h3), (h3, C)]

    def stage_in_a(self, s: int) -> int:
        """Width of GNNLayer input `a` at stage s (BS_brain.py:147 vs :154/:161)."""
        return self.node_in if s == 0 else self.feat_dim + self.node_in

    @property
    def params_per_slot(self) -> int:
        F, De = self.feat_dim, self.edge_in
        n = 0
        for s in range(self.n_mp_layers + 1):
            n += (self.stage_in_a(s) + De + F) * F + F
        for i, o in self.dense_dims:
            n += i * o + o
        return n

    @property
    def n_params(self) -> int:
        return self.params_per_slot * self.n_slots
