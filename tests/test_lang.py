"""Language auto-detection (TTSRequest(language="auto"); the reference uses langid, requests.py:96-113): every tag the
reference supports, on greeting-sized and paragraph-sized inputs."""
import pytest

from auralis_amd.api.lang import SUPPORTED, get_language, validate_language

SAMPLES = {
    "en": ["Hello there, how are you doing today my friend?",
           "The committee decided to postpone the meeting until next week because several members were ill.",
           "Please call me back."],
    "fr": ["Bonjour, comment allez-vous aujourd'hui mon ami ?",
           "Les enfants jouent dans le jardin pendant que leurs parents préparent le dîner.",
           "Merci beaucoup pour votre aide."],
    "de": ["Guten Tag, wie geht es Ihnen heute, mein Freund?",
           "Die Kinder spielen im Garten, während ihre Eltern das Abendessen vorbereiten.",
           "Vielen Dank für Ihre Hilfe."],
    "es": ["Hola, ¿cómo estás hoy, amigo mío? Espero que todo vaya bien.",
           "Los niños juegan en el jardín mientras sus padres preparan la cena.",
           "En un lugar de la Mancha, de cuyo nombre no quiero acordarme, vivía un hidalgo."],
    "it": ["Ciao, come stai oggi amico mio? Spero che tutto vada bene.",
           "I bambini giocano in giardino mentre i loro genitori preparano la cena.",
           "Nel mezzo del cammin di nostra vita mi ritrovai per una selva oscura."],
    "pt": ["Olá, como você está hoje, meu amigo? Espero que esteja tudo bem.",
           "As crianças brincam no jardim enquanto os pais preparam o jantar.",
           "Muito obrigado pela sua ajuda."],
    "pl": ["Cześć, jak się dzisiaj masz, przyjacielu?",
           "Dzieci bawią się w ogrodzie, podczas gdy rodzice przygotowują kolację."],
    "nl": ["Hallo, hoe gaat het vandaag met je, mijn vriend?",
           "De kinderen spelen in de tuin terwijl hun ouders het avondeten klaarmaken."],
    "tr": ["Merhaba, bugün nasılsın dostum?",
           "Çocuklar bahçede oynarken anne ve babaları akşam yemeğini hazırlıyor."],
    "cs": ["Ahoj, jak se dnes máš, příteli?",
           "Děti si hrají na zahradě, zatímco jejich rodiče připravují večeři."],
    "hu": ["Szia, hogy vagy ma, barátom?",
           "A gyerekek a kertben játszanak, miközben a szüleik vacsorát készítenek."],
    "ru": ["Привет, как у тебя дела сегодня, мой друг?"],
    "ja": ["こんにちは、今日はお元気ですか？"],
    "zh-cn": ["你好，你今天过得怎么样？"],
    "ko": ["안녕하세요, 오늘 어떻게 지내세요?"],
    "ar": ["مرحبا كيف حالك اليوم يا صديقي"],
    "hi": ["नमस्ते, आज आप कैसे हैं मेरे दोस्त?"],
}


@pytest.mark.parametrize("lang", sorted(SAMPLES))
def test_detects_every_supported_language(lang):
    for text in SAMPLES[lang]:
        assert get_language(text) == lang, text
    assert lang in SUPPORTED and validate_language(lang) == lang


def test_fallbacks_and_validation():
    assert get_language("") == "en" and get_language("12345 ... !!!") == "en"
    with pytest.raises(ValueError):
        validate_language("xx")
