"""CPU, build container only: the REFERENCE's own `Agent` (imported from /root/reference with the Keras/TF imports
stubbed) driving `v2xgnn.BS` through its real call sites -- `predict_one_step` in the rollout (BS_brain.py:336),
`predict` x 2 and `train_dnn` in `replay` (:664-665, :728), `update_target_model` (:847) -- with the dict payloads and
the dense kron(Adj, I_F) adjacency the reference builds.  The compute engine injected here is the float64 oracle (no GPU
in this container); the boundary code under test (BS / GnnQModel / packing) is the product's.
Skipped wherever /root/reference does not exist (the GPU box)."""
import os
import random
import sys

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout only exists in the build container")


def test_reference_agent_drives_v2xgnn_bs():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_golden as mg
    from oracle_engine import OracleEngine
    from oracle import literal as ol
    import v2xgnn
    saved = {k: sys.modules.get(k) for k in ('keras', 'keras.layers', 'keras.models', 'keras.backend',
                                               'keras.optimizers', 'tensorflow')}
    mg.install_stubs()
    sys.path.insert(0, REF)
    try:
        import Environment
        import Sim_Config
        import BS_brain
        random.seed(77)
        np.random.seed(77)
        cfg = Sim_Config.RL_Config()
        cfg.set_train_value(16, 0.5, 32, 1, 0.1)
        env = mg.make_env(Environment)
        made = []

        def make_bs(*args):                                   # the one-line swap of INTEGRATION.md section 1
            b = v2xgnn.BS(*args, seed=5, engine_factory=lambda spec: OracleEngine(spec))
            made.append(b)
            return b
        BS_brain.BS = make_bs
        BS_brain.Memory.samples = []
        agent = BS_brain.Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg)
        brain = made[0]
        assert agent.num_States == brain.num_D2D_Input == 4 * 13 + 16
        agent.num_Episodes, agent.num_Train_Step, agent.num_transition = 10, 20, 50
        agent.num_step = 10 ** 9                              # epsilon -> 0.01: the greedy branch calls predict_one_step
        w0 = np.concatenate([w.ravel() for w in brain.model.get_weights()])
        agent.generate_d2d_transition(24)                     # 24 < batch 32: the with-replacement sampling branch
        result, q_mean, q_max, _, _ = agent.replay()
        for k in range(1, 5):
            loss = result.history['D%d_Decide_Output_loss' % k][0]
            assert np.isfinite(loss) and loss >= 0
        assert len(q_mean) == 4 and np.all(q_max >= q_mean)
        w1 = np.concatenate([w.ravel() for w in brain.model.get_weights()])
        assert not np.array_equal(w0, w1)                     # train_dnn really stepped the online network
        # the reference's own state dict through BS.predict == the literal (dense kron) restatement on the same weights
        states = agent.generate_d2d_initial_states()
        got = brain.predict(states)
        from oracle import compact as oc
        from util import ospec
        osp = ospec(brain._spec)
        params = oc.params_from_list(osp, [np.asarray(w, np.float64) for w in brain.model.get_weights()], np.float64)
        ref = ol.forward_literal(osp, params, states)
        for a, b in zip(got, ref):
            assert a.shape == (1, 4) and np.allclose(a, b, rtol=2e-5, atol=2e-6)
        brain.update_target_model()
        for a, b in zip(brain.model.get_weights(), brain.target_model.get_weights()):
            assert np.array_equal(a, b)
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in ('Environment', 'Sim_Config', 'BS_brain'):
            sys.modules.pop(k, None)
