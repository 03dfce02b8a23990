"""``falcon_kit.mains``: only ``consensus`` is answered by the overlay; every other main
(``consensus_task``, ``run1`` ...) falls through to the reference package further down
sys.path."""
__path__ = __import__("pkgutil").extend_path(__path__, __name__)
