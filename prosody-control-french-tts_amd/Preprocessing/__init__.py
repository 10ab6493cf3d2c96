"""Drop-in mirrors of the reference's ``Code/Preprocessing`` feeders on the hot path."""
