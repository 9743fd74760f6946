"""Harness around the hot path: the reference-shaped VoteNet (backbone -> voting -> proposal),
its loss, a deterministic synthetic scene generator, and the one-process-per-GPU training step.
Counterpart of detection/Votenet/{models,train_Votenet_FSB.py} in the reference (SURVEY a-H)."""
