"""Bench and test scaffolding, not part of the product: synthetic parameter sets with the
reference models' initial distributions (``synth``) and a stand-in of the reference's model
classes that calls the drop-in operators in the reference's own sequence (``harness``)."""
