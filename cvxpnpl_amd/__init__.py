"""cvxpnpl_amd -- MI355X-native batched absolute-pose SDP solver.

Drop-in for the hot path of SergioRAgostinho/cvxpnpl (pnp / pnl / pnpl) plus batched
variants that solve tens of thousands to millions of independent problems per launch.
"""
__version__ = "0.1.0"

from .api import BatchResult, assemble_batch, pnl, pnl_batch, pnp, pnp_batch, pnpl, pnpl_batch, recover_multi, recover_multi_batch, score_hypotheses  # noqa: F401
