"""starkware.crypto.signature.signature, served by the MI355X implementation."""
from starkperp.signature import *  # noqa: F401,F403
from starkperp.signature import (  # noqa: F401  (names without __all__ semantics, explicit)
    ALPHA, BETA, CONSTANT_POINTS, EC_GEN, EC_ORDER, FIELD_GEN, FIELD_PRIME, MINUS_SHIFT_POINT,
    N_ELEMENT_BITS_ECDSA, N_ELEMENT_BITS_HASH, PEDERSEN_PARAMS, SHIFT_POINT, ECSignature,
    InvalidPublicKeyError, generate_k_rfc6979, get_random_private_key, get_y_coordinate, grind_key,
    inv_mod_curve_size, is_point_on_curve, is_valid_stark_key, is_valid_stark_private_key,
    mimic_ec_mult_air, pedersen_hash, pedersen_hash_as_point, private_key_to_ec_point_on_stark_curve,
    private_to_stark_key, sign, verify,
)
