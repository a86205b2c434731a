"""graph_weather/utils.py of the reference: coordinate validation."""
from graph_weather_amd.utils import validate_lat_lons  # noqa: F401
