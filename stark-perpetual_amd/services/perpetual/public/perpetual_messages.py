"""services.perpetual.public.perpetual_messages names, served by starkperp.perpetual_messages."""
from starkperp.perpetual_messages import (  # noqa: F401
    CONDITIONAL_TRANSFER, LIMIT_ORDER_WITH_FEES, TRANSFER, WITHDRAWAL, WITHDRAWAL_TO_ADDRESS, build_condition,
    get_conditional_transfer_msg, get_conditional_transfer_msg_without_bounds, get_limit_order_msg,
    get_limit_order_msg_without_bounds, get_price_msg, get_transfer_msg,
    get_transfer_msg_without_bounds, get_withdrawal_msg, get_withdrawal_msg_without_bounds,
    get_withdrawal_to_address_msg,
    get_withdrawal_to_address_msg_without_bounds, withdrawal_hash,
)
