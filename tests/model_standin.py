"""The stand-in containers live in the package (bench.py times the installed layers on them); kept importable under the old name."""
from stratified_transformer_amd.standin import *  # noqa: F401,F403
