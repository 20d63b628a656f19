"""Drop-in counterparts of the reference's ``gcn`` package (layers, models, utils) backed by libdgcn.so."""
