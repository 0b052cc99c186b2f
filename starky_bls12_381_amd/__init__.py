"""starky_bls12_381_amd: MI355X-native STARK prover for the BLS12-381 AIRs of Electron-Labs/starky_bls12_381.

Only the starky prove() hot path and what feeds it; see DESIGN.md.
"""
from .api import *  # noqa: F401,F403
