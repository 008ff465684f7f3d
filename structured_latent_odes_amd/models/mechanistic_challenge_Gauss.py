"""models/mechanistic_challenge_Gauss.py of the reference."""
from .mechanistic_challenge import MechanisticModel


class MechanisticModelGauss(MechanisticModel):
    GAUSS = True
