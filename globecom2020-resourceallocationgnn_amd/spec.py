"""Network sizes, mirroring `BS.__init__` of the reference (BS_brain.py:94-104)."""
from dataclasses import dataclass

XE_WIDTH = 16            # packed [node | edge | pad] row (include/v2xgnn.h V2X_XE_WIDTH)
HIDDEN = (80, 40, 20)    # Dense widths, BS_brain.py:176-178


@dataclass(frozen=True)
class GnnSpec:
    n_nodes: int = 4           # num_D2D                      (BS_brain.py:95)
    n_channels: int = 4        # num_CH                       (:97)
    feat_dim: int = 16         # num_Feedback                 (:98)
    n_mp_layers: int = 2       # GNN stages after the embed   (:154-164)
    share_weights: bool = False  # reference keeps one weight set per node (:121-200)
    variable_graphs: bool = False
    input_node_info: int = 3   # (:294)
    input_edge_info: int = 1   # (:295)
    n_neighbor: int = 1        # (:96)

    @property
    def node_in(self):   # num_One_Node_Input (:101)
        return ((self.input_node_info - 1) * self.n_channels + 1) * self.n_neighbor

    @property
    def edge_in(self):   # num_One_Edge_Input (:102)
        return self.input_edge_info * self.n_channels

    @property
    def n_slots(self):
        return 1 if self.share_weights else self.n_nodes

    def stage_in_a(self, s):
        return self.node_in if s == 0 else self.feat_dim + self.node_in

    @property
    def dense_dims(self):
        F, Dn, C = self.feat_dim, self.node_in, self.n_channels
        return [(Dn + 2 * F, HIDDEN[0]), (HIDDEN[0], HIDDEN[1]), (HIDDEN[1], HIDDEN[2]), (HIDDEN[2], C)]

    @property
    def params_per_slot(self):
        F, De = self.feat_dim, self.edge_in
        n = sum((self.stage_in_a(s) + De + F) * F + F for s in range(self.n_mp_layers + 1))
        return n + sum(i * o + o for i, o in self.dense_dims)

    @property
    def n_params(self):
        return self.params_per_slot * self.n_slots
