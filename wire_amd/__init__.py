"""wire_amd -- the WIRE INR hot path on MI355X (gfx950).

Public surface mirrors the reference's ``modules`` package:
    from wire_amd.modules import models, utils
    model = models.get_INR(nonlin='wire', in_features=2, out_features=3,
                           hidden_features=256, hidden_layers=4, ...).cuda()
Arithmetic lives in wire_amd/lib/libwire_hip.so (include/wire_hip.h).
"""
__version__ = "0.1.0"
