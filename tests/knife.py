"""Knife-edge allowances say what they let through.  Several parity tests accept that ONE chain (a few particles) may differ from the oracle
where an accept test or a resampling threshold sits within rounding of its boundary.  Every use of such an allowance is reported as a
warning -- pytest lists warnings at the end of a run whatever its verbosity -- with the test, the chains and what was compared, so that a
systematic one-chain bug cannot hide behind it: the same chain in every run, or a distance that is not small, shows."""
import warnings


class KnifeEdge(UserWarning):
    pass


def used(where, **what):
    """Report a use of a knife-edge allowance.  where: the comparison; what: chain / step indices, the distance to the boundary, ..."""
    warnings.warn(KnifeEdge(f"knife-edge allowance used in {where}: " + ", ".join(f"{k}={v}" for k, v in what.items())), stacklevel=2)
