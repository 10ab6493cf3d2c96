"""Drop-in mirrors of the reference's ``Code/Pipeline`` modules (same module and function names).
Put this package's directory on ``sys.path`` in place of ``Code/`` and the legacy imports
(``from Pipeline.compute_rate_adjustments import calculate_rate`` ...) resolve here."""
