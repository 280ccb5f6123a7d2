"""Model factory -- the drop-in boundary (reference modules/models.py:15-77).

``get_INR`` keeps the reference's name, positional order and keyword names, and
is a superset of it: at the reference commit ``scaled_hidden_features`` is a
required positional that none of the wire_*.py drivers passes
(wire_image_denoise.py:106-118, wire_occupancy.py:107-116 -> TypeError), and
the 14 positionals are forwarded blindly to INR classes that take 12
(SURVEY.md fact 3).  Here the argument is optional and each module receives the
arguments its own signature has, so both call styles work:

    get_INR(nonlin='wire', in_features=2, out_features=3, hidden_features=256,
            hidden_layers=2, first_omega_0=7., hidden_omega_0=7., scale=6.)
    get_INR('wire', 2, 300, 0, 2, 3, scale_tensor=[0.0], ...)   # bspline_* style
"""
from . import gauss, relu, siren, wire, wire2d

# keys of modules/models.py:15-25 that are on the MI355X path; 'mfn' and the
# bspline_* family are out of scope.
model_dict = {'gauss': gauss,
              'relu': relu,
              'siren': siren,
              'wire': wire,
              'wire2d': wire2d}

_OUT_OF_SCOPE = ('mfn', 'bspline_form', 'bspline_cubic', 'bspline_mscale_2',
                 'bspline_mscale_HL', 'bspline_mscale_hier')


def get_INR(nonlin, in_features, hidden_features, scaled_hidden_features=None,
            hidden_layers=None, out_features=None, outermost_linear=True,
            first_omega_0=30, hidden_omega_0=30, scale=10, scale_tensor=[],
            pos_encode=False, sidelength=512, fn_samples=None, use_nyquist=True):
    """Return an INR ``nn.Module`` whose forward/backward run on MI355X.

    nonlin: 'wire', 'wire2d', 'siren', 'gauss' or 'relu' ('posenc' is 'relu'
    with ``pos_encode=True``, as the reference's drivers spell it).
    Remaining arguments: see modules/models.py:31-56 of the reference.
    """
    if nonlin in _OUT_OF_SCOPE:
        raise NotImplementedError(f"nonlin '{nonlin}' is outside the MI355X hot path of wire_amd")
    if nonlin not in model_dict:
        raise KeyError(nonlin)
    if hidden_layers is None or out_features is None:
        raise TypeError("get_INR() needs hidden_layers and out_features")
    mod = model_dict[nonlin]
    if nonlin == 'wire':
        # 15-argument form, modules/wire.py:96-111
        return mod.INR(in_features, hidden_features,
                       0 if scaled_hidden_features is None else scaled_hidden_features,
                       hidden_layers, out_features, outermost_linear, first_omega_0,
                       hidden_omega_0, scale, scale_tensor, pos_encode,
                       sidelength=sidelength, fn_samples=fn_samples, use_nyquist=use_nyquist)
    # 12-argument form shared by siren / gauss / relu / wire2d
    return mod.INR(in_features, hidden_features, hidden_layers, out_features, outermost_linear,
                   first_omega_0, hidden_omega_0, scale, pos_encode, sidelength, fn_samples,
                   use_nyquist)
