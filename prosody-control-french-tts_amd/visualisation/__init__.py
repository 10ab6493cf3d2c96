"""Drop-in mirrors of the array functions of the reference's viewers (``Code/visualisation``)."""
