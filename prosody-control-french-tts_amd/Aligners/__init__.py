"""Drop-in mirrors of the reference's ``Code/Aligners`` modules on the hot path."""
