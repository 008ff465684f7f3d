"""Hyper-parameters of the three reference configurations (data/cvs/config_cvs.py:6-52, data/proc/config_proc.py:9-66,
data/challenge/config_challenge.py:6-49) as attribute dictionaries (the reference uses `munch`, absent here).  Values are
transcribed; dataset paths / fold bookkeeping (out of scope) are omitted."""


class AttrDict(dict):
    """dict with attribute access (stand-in for munch.Munch)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def copy(self):
        return AttrDict(self)


def _common():
    return AttrDict(u_hidden_dim=25, aux_loss_multiplier=46.0, seed=12, n_filters=10, filter_size=10, pool_size=5,
                    cnn_hidden_dim=50, ode_hidden_dim=25, num_particles=1, adjoint_solver=True, solver="midpoint",
                    constant_std=1e-2, quantile_diff=0.475, model="Mechanistic")


def load_config_cvs():
    c = _common()
    c.update(seq_len=86, data_size=1000, delta_t=1.0, norm="zero_to_one", obs_dim=3, iext_dim=1, rtpr_dim=1, z_iext_dim=5,
             z_rtpr_dim=5, z_epsilon_dim=5, num_epochs=1000, plot_epoch=100, mini_batch_size=128, ode_state_dim=5,
             system_input_dim=2, learning_rate=1e-3)
    return c


def load_config_challenge():
    c = _common()
    c.update(seq_len=142, delta_t=1.0, norm="zero_to_one", obs_dim=4, shedding_dim=1, symptoms_dim=1, z_shedding_dim=5,
             z_symptoms_dim=5, z_epsilon_dim=5, num_epochs=500, plot_epoch=250, mini_batch_size=100, folds=5, split=5,
             ode_state_dim=5, system_input_dim=2, learning_rate=1e-3, num_samples=200)
    return c


def load_config_proc():
    c = _common()
    c.update(seq_len=86, obs_dim=4, aR_dim=3, aS_dim=4, C12_dim=1, C6_dim=1, z_aR_dim=10, z_aS_dim=10, z_C12_dim=10,
             z_C6_dim=10, z_epsilon_dim=10, num_epochs=2500, plot_epoch=200, mini_batch_size=36, ode_state_dim=8,
             system_input_dim=9, learning_rate=3e-4, num_samples=200, heldout=None, folds=4, split=1)
    return c
