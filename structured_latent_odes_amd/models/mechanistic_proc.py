"""models/mechanistic_proc.py of the reference (kwargs ``observations, aR, aS, C12, C6``; prior input cat(aR, aS, C12, C6),
mechanistic_proc.py:196-198; latent layout [z_aR, z_aS, z_C12, z_C6, z_epsilon], :282-311).

The proc MAIN loss additionally scores the labels on the replayed z at 46x (mechanistic_proc.py:145-146,334-353); those
four heads (two softmax/OneHotCategorical, two Exp/Laplace) are evaluated and differentiated inside ``ode_elbo_kernel``
(phases P0/P7), so their parameters live in the kernel's layout for this family."""
from ._mechanistic import MechanisticBase


class MechanisticModel(MechanisticBase):
    FAMILY, GAUSS = "proc", False
    LABELS = ("aR", "aS", "C12", "C6")
    Z_GROUPS = ("aR", "aS", "C12", "C6", "epsilon")
    PRIORS = [("p_z_u_given_u", ["aR", "aS", "C12", "C6"], ["aR", "aS", "C12", "C6"])]
    AUX = [("q_aR_given_z_aR", "aR", "aR", "softmax"), ("q_aS_given_z_aS", "aS", "aS", "softmax"),
           ("q_C12_given_z_C12", "C12", "C12", "expexp"), ("q_C6_given_z_C6", "C6", "C6", "expexp")]
    LABELS_IN_MAIN = True

    def pred_inputs(self, observations):
        """Predicted aR, aS (one-hot), C12, C6 (mechanistic_proc.py:361-392)."""
        return self._predict_labels(observations)
