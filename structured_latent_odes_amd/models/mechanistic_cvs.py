"""models/mechanistic_cvs.py of the reference: ``MechanisticModel(config, device, times)`` for the CVS data set
(kwargs ``observations, iext, rtpr``; attribute names as in mechanistic_cvs.py:58-103)."""
from ._mechanistic import MechanisticBase


class MechanisticModel(MechanisticBase):
    FAMILY, GAUSS = "cvs", False
    LABELS = ("iext", "rtpr")
    Z_GROUPS = ("iext", "rtpr", "epsilon")
    PRIORS = [("p_z_iext_given_iext", ["iext"], ["iext"]), ("p_z_rtprs_given_rtprs", ["rtpr"], ["rtpr"])]
    AUX = [("q_iext_given_z_iext", "iext", "iext", "sigmoid"), ("q_rtpr_given_z_rtpr", "rtpr", "rtpr", "sigmoid")]

    def classifier(self, observations):
        """Predicted iext / rtpr (mechanistic_cvs.py:278-296)."""
        return self._predict_labels(observations)
