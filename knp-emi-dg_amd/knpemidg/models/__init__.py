"""Membrane (ODE) models in the vectorised module protocol of knpemidg.membrane."""
