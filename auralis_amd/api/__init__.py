"""Host-side mirror of the reference's Python surface for the generate_speech() path
(TTS / TTSRequest / TTSOutput / BaseAsyncTTSEngine / registry / two-phase scheduler)."""
