"""models/mechanistic_proc_Gauss.py of the reference."""
from .mechanistic_proc import MechanisticModel


class MechanisticModelGauss(MechanisticModel):
    GAUSS = True
