"""models/mechanistic_challenge.py of the reference (kwargs ``observations, symptoms, shedding``; prior input is
cat(symptoms, shedding), mechanistic_challenge.py:167; latent layout [z_shedding, z_symptoms, z_epsilon], :244-262)."""
from ._mechanistic import MechanisticBase


class MechanisticModel(MechanisticBase):
    FAMILY, GAUSS = "challenge", False
    LABELS = ("symptoms", "shedding")
    Z_GROUPS = ("shedding", "symptoms", "epsilon")
    PRIORS = [("p_z_u_given_u", ["symptoms", "shedding"], ["shedding", "symptoms"])]
    AUX = [("q_shedding_given_z_shedding", "shedding", "shedding", "sigmoid"),
           ("q_symptom_given_z_symptom", "symptoms", "symptoms", "sigmoid")]

    def pred_inputs(self, observations):
        """Predicted shedding / symptoms (mechanistic_challenge.py:299-313)."""
        return self._predict_labels(observations)
