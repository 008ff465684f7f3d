"""models/mechanistic_cvs_Gauss.py of the reference: the Gaussian-likelihood ablation (``MechanisticModelGauss``)."""
from .mechanistic_cvs import MechanisticModel


class MechanisticModelGauss(MechanisticModel):
    GAUSS = True
