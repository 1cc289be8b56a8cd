"""cvxpnpl_amd -- MI355X-native batched absolute-pose SDP solver.

Drop-in for the hot path of SergioRAgostinho/cvxpnpl (pnp / pnl / pnpl) plus batched
variants that solve tens of thousands to millions of independent problems per launch.
"""
__version__ = "0.1.0"

from ._lib import VARIANT_FULL, VARIANT_RC  # noqa: F401
from .api import (BatchResult, assemble_batch, assemble_subsets, ipm_batch, pack_cost, pnl, pnl_batch, pnp, pnp_batch, pnpl, pnpl_batch, recover_multi,  # noqa: F401
                  recover_multi_batch, recover_multi_device, refit_update, sample_minimal_sets, score_hypotheses, select_best, solve_cost_batch, solve_relaxation, solve_relaxation_rc)
