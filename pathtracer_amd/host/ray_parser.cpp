// ray_parser.cpp — reader for the ".ray" scene text format (SBT-raytracer 1.x).
//
// Behavioural counterpart of the reference's Tokenizer/Parser
// (src/parser/Tokenizer.cu, src/parser/Token.cu:112-177, src/parser/Parser.cu);
// own recursive-descent implementation over an in-memory string.  Grammar quirks
// of the reference that change the scene are kept and cited inline.
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <vector>

#include "scene.h"

namespace pt {
namespace {

enum Tok {
    T_EOF, T_IDENT, T_SCALAR, T_LPAREN, T_RPAREN, T_LBRACE, T_RBRACE, T_COMMA, T_EQUALS, T_SEMI,
    // reserved words (Token.cu:117-166)
    K_SBT, K_AMBIENT_LIGHT, K_AMBIENT, K_ASPECT, K_BOTTOM_RADIUS, K_BOX, K_CAMERA, K_CAPPED, K_COLOR, K_CONE,
    K_CONST_ATT, K_CYLINDER, K_DIFFUSE, K_DIRECTION, K_DIR_LIGHT, K_EMISSIVE, K_FACES, K_FALSE, K_FOV, K_GENNORMALS,
    K_HEIGHT, K_INDEX, K_LIN_ATT, K_MATERIAL, K_MATERIALS, K_MAP, K_NAME, K_NORMALS, K_POINT_LIGHT, K_POINTS,
    K_TRIMESH, K_POSITION, K_QUAD_ATT, K_QUATERNION, K_REFLECTIVE, K_ROTATE, K_SCALE, K_SHININESS, K_SPECULAR,
    K_SPHERE, K_SQUARE, K_TOP_RADIUS, K_TRANSFORM, K_TRANSLATE, K_TRANSMISSIVE, K_TRUE, K_UPDIR, K_VIEWDIR
};

const std::map<std::string, Tok> &reserved() {
    static const std::map<std::string, Tok> r = {
        {"ambient_light", K_AMBIENT_LIGHT}, {"ambient", K_AMBIENT}, {"aspectratio", K_ASPECT},
        {"bottom_radius", K_BOTTOM_RADIUS}, {"box", K_BOX}, {"camera", K_CAMERA}, {"capped", K_CAPPED},
        {"color", K_COLOR}, {"colour", K_COLOR}, {"cone", K_CONE}, {"constant_attenuation_coeff", K_CONST_ATT},
        {"cylinder", K_CYLINDER}, {"diffuse", K_DIFFUSE}, {"direction", K_DIRECTION},
        {"directional_light", K_DIR_LIGHT}, {"emissive", K_EMISSIVE}, {"faces", K_FACES}, {"false", K_FALSE},
        {"fov", K_FOV}, {"gennormals", K_GENNORMALS}, {"height", K_HEIGHT}, {"index", K_INDEX},
        {"linear_attenuation_coeff", K_LIN_ATT}, {"material", K_MATERIAL}, {"materials", K_MATERIALS},
        {"map", K_MAP}, {"name", K_NAME}, {"normals", K_NORMALS}, {"point_light", K_POINT_LIGHT},
        {"points", K_POINTS}, {"polymesh", K_TRIMESH}, {"position", K_POSITION},
        {"quadratic_attenuation_coeff", K_QUAD_ATT}, {"quaternian", K_QUATERNION}, {"reflective", K_REFLECTIVE},
        {"rotate", K_ROTATE}, {"SBT-raytracer", K_SBT}, {"scale", K_SCALE}, {"shininess", K_SHININESS},
        {"specular", K_SPECULAR}, {"sphere", K_SPHERE}, {"square", K_SQUARE}, {"top_radius", K_TOP_RADIUS},
        {"transform", K_TRANSFORM}, {"translate", K_TRANSLATE}, {"transmissive", K_TRANSMISSIVE},
        {"trimesh", K_TRIMESH}, {"true", K_TRUE}, {"updir", K_UPDIR}, {"viewdir", K_VIEWDIR}};
    return r;
}

struct Token {
    Tok kind = T_EOF;
    double value = 0;
    std::string ident;
};

class Lexer {
public:
    explicit Lexer(const std::string &text) : s(text) {}

    [[noreturn]] void fail(const std::string &msg) const {
        std::ostringstream o; // ParserException.cu:20-21
        o << "Line " << line << ": syntax error: " << msg;
        throw ParseError{o.str()};
    }
    const Token &peek() {
        if (!havePeek) { peeked = next(); havePeek = true; }
        return peeked;
    }
    Token get() {
        if (havePeek) { havePeek = false; return peeked; }
        return next();
    }
    Token read(Tok kind, const char *what) { // Tokenizer.cu:312-320
        Token t = get();
        if (t.kind != kind) fail(std::string(what) + " expected");
        return t;
    }
    bool condRead(Tok kind) { // Tokenizer.cu:331-339
        if (peek().kind == kind) { get(); return true; }
        return false;
    }

private:
    char cur() const { return pos < s.size() ? s[pos] : '\0'; }
    void adv() {
        if (pos < s.size()) { if (s[pos] == '\n') line++; pos++; }
    }
    void skipSpace() { // Tokenizer.cu:130-174
        for (;;) {
            while (pos < s.size() && isspace((unsigned char)cur())) adv();
            if (cur() != '/') return;
            adv();
            if (cur() == '/') {
                while (pos < s.size() && cur() != '\n') adv();
            } else if (cur() == '*') {
                int start = line;
                adv();
                for (;;) {
                    if (pos >= s.size()) {
                        std::ostringstream o;
                        o << "Unterminated comment in line " << start;
                        fail(o.str());
                    }
                    if (cur() == '*') { adv(); if (cur() == '/') { adv(); break; } }
                    else adv();
                }
            } else {
                fail(std::string("unexpected character: '") + cur() + "'");
            }
        }
    }
    Token next() {
        skipSpace();
        Token t;
        if (pos >= s.size()) return t;
        char c = cur();
        if (isalpha((unsigned char)c) || c == '_') { // Tokenizer.cu:199-208: '-' continues an identifier
            std::string id;
            while (isalnum((unsigned char)cur()) || cur() == '_' || cur() == '-') { id += cur(); adv(); }
            auto it = reserved().find(id);
            if (it != reserved().end()) t.kind = it->second;
            else { t.kind = T_IDENT; t.ident = id; }
        } else if (c == '"') { // Tokenizer.cu:176-189
            adv();
            while (cur() != '"') {
                if (cur() == '\n' || pos >= s.size()) fail("Unterminated string constant");
                t.ident += cur();
                adv();
            }
            adv();
            t.kind = T_IDENT;
        } else if (isdigit((unsigned char)c) || c == '-' || c == '.') { // Tokenizer.cu:217-225: [0-9.\-e]* then atof
            std::string num;
            while (isdigit((unsigned char)cur()) || cur() == '-' || cur() == '.' || cur() == 'e') { num += cur(); adv(); }
            t.kind = T_SCALAR;
            t.value = atof(num.c_str());
        } else {
            switch (c) { // Tokenizer.cu:234-274
                case '(': t.kind = T_LPAREN; break;
                case ')': t.kind = T_RPAREN; break;
                case '{': t.kind = T_LBRACE; break;
                case '}': t.kind = T_RBRACE; break;
                case ',': t.kind = T_COMMA; break;
                case '=': t.kind = T_EQUALS; break;
                case ';': t.kind = T_SEMI; break;
                default: fail(std::string("unexpected character: '") + c + "'");
            }
            adv();
        }
        return t;
    }

    const std::string &s;
    size_t pos = 0;
    int line = 1;
    bool havePeek = false;
    Token peeked;
};

inline V4 xformPoint(const M4 &m, float x, float y, float z) { // transform.h:3-7,22-24
    V4 r = m * V4(x, y, z, 1.0f);
    return V4(r.x, r.y, r.z);
}

bool isTransformable(Tok k) {
    switch (k) {
        case K_SPHERE: case K_BOX: case K_SQUARE: case K_CYLINDER: case K_CONE: case K_TRIMESH:
        case K_TRANSLATE: case K_ROTATE: case K_SCALE: case K_TRANSFORM: case T_LBRACE: return true;
        default: return false;
    }
}

class Parser {
public:
    Parser(const std::string &text, const std::string &base) : lx(text), basePath(base) {}

    Scene *parseScene() { // Parser.cu:20-78
        std::unique_ptr<Scene> sc(new Scene);
        scene = sc.get();
        lx.read(K_SBT, "SBT-raytracer");
        Token ver = lx.read(T_SCALAR, "scalar");
        if (ver.value > 1.1) {
            std::ostringstream o;
            o << "SBT-raytracer version number " << ver.value << " too high; only able to parse v1.1 and below.";
            throw ParseError{o.str()};
        }
        Material mat;
        M4 root = M4::identity();
        for (;;) {
            Tok k = lx.peek().kind;
            if (isTransformable(k)) parseTransformableElement(root, mat);
            else if (k == K_POINT_LIGHT) parsePointLight();
            else if (k == K_DIR_LIGHT) parseDirectionalLight();
            else if (k == K_AMBIENT_LIGHT) parseAmbientLight();
            else if (k == K_CAMERA) parseCamera();
            else if (k == K_MATERIAL) mat = parseMaterialExpression(mat);
            else if (k == T_SEMI) lx.get();
            else if (k == T_EOF) return sc.release();
            else lx.fail("Expected: geometry, camera, or light information");
        }
    }

private:
    float scalar() { return (float)lx.read(T_SCALAR, "scalar").value; } // Parser.cu:733-736
    V4 vec3() { // Parser.cu:777-786
        lx.read(T_LPAREN, "(");
        float a = scalar(); lx.read(T_COMMA, ",");
        float b = scalar(); lx.read(T_COMMA, ",");
        float c = scalar(); lx.read(T_RPAREN, ")");
        return V4(a, b, c);
    }
    V4 vec4Dropped() { // Parser.cu:788-799: the 4th component is read and DROPPED
        lx.read(T_LPAREN, "(");
        float a = scalar(); lx.read(T_COMMA, ",");
        float b = scalar(); lx.read(T_COMMA, ",");
        float c = scalar(); lx.read(T_COMMA, ",");
        scalar(); lx.read(T_RPAREN, ")");
        return V4(a, b, c);
    }
    void vec4Full(float out[4]) { // rows of `transform(...)`: Parser.cu:299-313 reads row[3] of an f4 built
        V4 v = vec4Dropped();     // by parseVec4d, i.e. 0 — the translation column of the matrix is lost.
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    }
    float scalarExpr() { lx.get(); lx.read(T_EQUALS, "="); float v = scalar(); lx.condRead(T_SEMI); return v; } // :684-691
    V4 vec3Expr() { lx.get(); lx.read(T_EQUALS, "="); V4 v = vec3(); lx.condRead(T_SEMI); return v; }          // :701-707
    V4 vec4Expr() { lx.get(); lx.read(T_EQUALS, "="); V4 v = vec4Dropped(); lx.condRead(T_SEMI); return v; }   // :709-715
    void identExpr() { lx.get(); lx.read(T_EQUALS, "="); lx.read(T_IDENT, "identifier"); lx.condRead(T_SEMI); } // :725-731
    void boolExpr() { // :693-699,764-775
        lx.get(); lx.read(T_EQUALS, "=");
        Tok k = lx.peek().kind;
        if (k == K_TRUE || k == K_FALSE) lx.get(); else lx.fail("Expected boolean");
        lx.condRead(T_SEMI);
    }

    void parseCamera() { // Parser.cu:80-130
        bool hasView = false, hasUp = false;
        V4 viewDir, upDir;
        lx.read(K_CAMERA, "camera");
        lx.read(T_LBRACE, "{");
        for (;;) {
            switch (lx.peek().kind) {
                case K_POSITION: scene->camera.setEye(vec3Expr()); break;
                case K_FOV: scene->camera.setFOV(scalarExpr()); break;
                case K_QUATERNION: { V4 q = vec4Expr(); scene->camera.setLookQuat(q.x, q.y, q.z, q.w); break; }
                case K_ASPECT: scene->camera.setAspectRatio(scalarExpr()); break;
                case K_VIEWDIR: viewDir = vec3Expr(); hasView = true; break;
                case K_UPDIR: upDir = vec3Expr(); hasUp = true; break;
                case T_RBRACE:
                    if (hasView) {
                        if (!hasUp) lx.fail("Expected: 'updir'");
                        scene->camera.setLook(viewDir, upDir);
                    } else if (hasUp) {
                        lx.fail("Expected: 'viewdir'");
                    }
                    lx.get();
                    return;
                default: lx.fail("Expected: camera attribute");
            }
        }
    }

    void parseTransformableElement(const M4 &xf, const Material &mat) { // Parser.cu:132-153
        Tok k = lx.peek().kind;
        if (k == T_LBRACE) parseGroup(xf, mat);
        else if (isTransformable(k)) parseGeometry(xf, mat);
        else lx.fail("Expected: transformable element");
    }

    void parseGroup(const M4 &xf, const Material &mat) { // Parser.cu:156-186
        lx.read(T_LBRACE, "{");
        for (;;) {
            Tok k = lx.peek().kind;
            if (isTransformable(k)) parseTransformableElement(xf, mat);
            else if (k == T_RBRACE) { lx.get(); return; }
            else if (k == K_MATERIAL) {
                // :178-183: the MATERIAL case has no `break` and falls into the
                // default branch, so a material inside a group is a syntax error.
                parseMaterialExpression(mat);
                lx.fail("Expected: '}' or geometry");
            } else lx.fail("Expected: '}' or geometry");
        }
    }

    void parseGeometry(const M4 &xf, const Material &mat) { // Parser.cu:189-225
        switch (lx.peek().kind) {
            case K_SPHERE: parseSphere(xf, mat); return;
            case K_BOX: parseBoxLike(xf, mat, K_BOX); return;
            case K_SQUARE: parseBoxLike(xf, mat, K_SQUARE); return;
            case K_CYLINDER: parseIgnoredSolid(K_CYLINDER, mat); return;
            case K_CONE: parseIgnoredSolid(K_CONE, mat); return;
            case K_TRIMESH: parseTrimesh(xf, mat); return;
            case K_TRANSLATE: { // :228-244
                lx.get(); lx.read(T_LPAREN, "(");
                float x = scalar(); lx.read(T_COMMA, ",");
                float y = scalar(); lx.read(T_COMMA, ",");
                float z = scalar(); lx.read(T_COMMA, ",");
                parseTransformableElement(xf * translate(x, y, z), mat);
                lx.read(T_RPAREN, ")"); lx.condRead(T_SEMI);
                return;
            }
            case K_ROTATE: { // :246-264
                lx.get(); lx.read(T_LPAREN, "(");
                float x = scalar(); lx.read(T_COMMA, ",");
                float y = scalar(); lx.read(T_COMMA, ",");
                float z = scalar(); lx.read(T_COMMA, ",");
                float w = scalar(); lx.read(T_COMMA, ",");
                parseTransformableElement(xf * rotate(w, x, y, z), mat);
                lx.read(T_RPAREN, ")"); lx.condRead(T_SEMI);
                return;
            }
            case K_SCALE: { // :267-292
                lx.get(); lx.read(T_LPAREN, "(");
                float x = scalar(), y, z;
                lx.read(T_COMMA, ",");
                if (lx.peek().kind == T_SCALAR) {
                    y = scalar(); lx.read(T_COMMA, ",");
                    z = scalar(); lx.read(T_COMMA, ",");
                } else { y = x; z = x; }
                parseTransformableElement(xf * scale(x, y, z), mat);
                lx.read(T_RPAREN, ")"); lx.condRead(T_SEMI);
                return;
            }
            case K_TRANSFORM: { // :295-317
                lx.get(); lx.read(T_LPAREN, "(");
                float r[4][4];
                for (int i = 0; i < 4; i++) { vec4Full(r[i]); lx.read(T_COMMA, ","); }
                M4 m;
                for (int i = 0; i < 4; i++) m.c[i] = V4(r[i][0], r[i][1], r[i][2], r[i][3]);
                parseTransformableElement(xf * transpose(m), mat);
                lx.read(T_RPAREN, ")"); lx.condRead(T_SEMI);
                return;
            }
            default: throw ParseError{"Unrecognized geometry type."};
        }
    }

    void parseSphere(const M4 &xf, const Material &mat) { // Parser.cu:319-346
        Material newMat = mat;
        lx.read(K_SPHERE, "sphere"); lx.read(T_LBRACE, "{");
        for (;;) {
            switch (lx.peek().kind) {
                case K_MATERIAL: newMat = parseMaterialExpression(mat); break;
                case K_NAME: identExpr(); break;
                case T_RBRACE: {
                    lx.get();
                    V4 c = xformPoint(xf, 0, 0, 0);                 // :339
                    float radius = (xf * V4(1, 0, 0, 0)).x;         // :340 — radius is xform[0][0] only
                    scene->addSphere(c, radius, scene->addMaterial(newMat));
                    return;
                }
                default: lx.fail("Expected: sphere attributes");
            }
        }
    }

    void addCube(const V4 v[8], const Material &mat) { // trimesh.cu:131-147
        static const int idx[12][3] = {{0, 1, 2}, {3, 2, 1}, {4, 6, 5}, {7, 5, 6}, {0, 2, 4}, {6, 4, 2},
                                       {1, 5, 3}, {7, 3, 5}, {0, 1, 4}, {5, 4, 1}, {2, 6, 3}, {7, 3, 6}};
        uint32_t m = scene->addMaterial(mat);
        uint32_t mm[3] = {m, m, m};
        for (auto &f : idx) scene->addTriangle(v[f[0]], v[f[1]], v[f[2]], mm, false);
    }

    void parseBoxLike(const M4 &xf, const Material &mat, Tok kind) { // Parser.cu:348-398
        Material newMat = mat;
        lx.get(); lx.read(T_LBRACE, "{");
        for (;;) {
            switch (lx.peek().kind) {
                case K_MATERIAL: newMat = parseMaterialExpression(mat); break;
                case K_NAME: identExpr(); break;
                case T_RBRACE: {
                    lx.get();
                    // trimesh.cu:149-173: unit cube +-0.5, or a 0.002-thick slab for `square`
                    const float hz = kind == K_BOX ? 0.5f : (float)0.001;
                    V4 v[8];
                    int q = 0;
                    for (int ix = 0; ix < 2; ix++)
                        for (int iy = 0; iy < 2; iy++)
                            for (int iz = 0; iz < 2; iz++)
                                v[q++] = xformPoint(xf, ix ? 0.5f : -0.5f, iy ? 0.5f : -0.5f, iz ? hz : -hz);
                    addCube(v, newMat);
                    return;
                }
                default: lx.fail(kind == K_BOX ? "Expected: box attributes" : "Expected: square attributes");
            }
        }
    }

    void parseIgnoredSolid(Tok kind, const Material &mat) { // Parser.cu:400-450: cylinder/cone are parsed and discarded
        lx.get(); lx.read(T_LBRACE, "{");
        for (;;) {
            Tok k = lx.peek().kind;
            if (k == K_MATERIAL) parseMaterialExpression(kind == K_CYLINDER ? Material() : mat);
            else if (k == K_NAME) identExpr();
            else if (kind == K_CONE && k == K_CAPPED) boolExpr();
            else if (kind == K_CONE && (k == K_BOTTOM_RADIUS || k == K_TOP_RADIUS || k == K_HEIGHT)) scalarExpr();
            else if (k == T_RBRACE) { lx.get(); return; }
            else lx.fail(kind == K_CYLINDER ? "Expected: cylinder attributes" : "Expected: cone attributes");
        }
    }

    void parseFaces(std::list<V4> &faces) { // Parser.cu:580-596, fan triangulation
        std::vector<float> pts;
        lx.read(T_LPAREN, "(");
        if (lx.peek().kind != T_RPAREN) {
            pts.push_back(scalar());
            while (lx.peek().kind != T_RPAREN) { lx.read(T_COMMA, ","); pts.push_back(scalar()); }
        }
        lx.read(T_RPAREN, ")");
        if (pts.size() < 3) lx.fail("Faces must have at least 3 vertices.");
        float a = pts[0], b = pts[1];
        for (size_t i = 2; i < pts.size(); i++) { faces.emplace_back(a, b, pts[i]); b = pts[i]; }
    }

    template <class F>
    void parenList(F item) { // "( item, item, ... );"
        lx.get(); lx.read(T_EQUALS, "="); lx.read(T_LPAREN, "(");
        if (lx.peek().kind != T_RPAREN) {
            item();
            while (lx.peek().kind != T_RPAREN) { lx.read(T_COMMA, ","); item(); }
        }
        lx.read(T_RPAREN, ")"); lx.read(T_SEMI, ";");
    }

    void parseTrimesh(const M4 &xf, const Material &mat) { // Parser.cu:452-578
        Material meshMat = mat;
        std::vector<V4> verts;
        std::vector<Material> vmats;
        std::list<V4> faces;
        lx.read(K_TRIMESH, "trimesh"); lx.read(T_LBRACE, "{");
        for (;;) {
            switch (lx.peek().kind) {
                case K_GENNORMALS: lx.get(); lx.read(T_SEMI, ";"); break;
                case K_MATERIAL: meshMat = parseMaterialExpression(mat); break;
                case K_NAME: identExpr(); break;
                case K_MATERIALS: parenList([&] { vmats.push_back(parseMaterial(meshMat)); }); break;
                case K_NORMALS: parenList([&] { vec3(); }); break; // read and ignored (:496-512)
                case K_FACES: parenList([&] { parseFaces(faces); }); break;
                case K_POINTS: parenList([&] { V4 v = vec3(); verts.push_back(xformPoint(xf, v.x, v.y, v.z)); }); break;
                case T_RBRACE: {
                    lx.get();
                    std::vector<uint32_t> vm(vmats.size());
                    for (size_t i = 0; i < vmats.size(); i++) vm[i] = scene->addMaterial(vmats[i]);
                    uint32_t mm = scene->addMaterial(meshMat);
                    const int vCount = (int)verts.size();
                    for (const V4 &f : faces) {
                        int a = (int)f.x, b = (int)f.y, c = (int)f.z;
                        if (a >= vCount || b >= vCount || c >= vCount || a < 0 || b < 0 || c < 0) { // trimesh.cu:63-67
                            std::ostringstream o;
                            o << "Bad face in trimesh: (" << f.x << ", " << f.y << ", " << f.z << ")";
                            throw ParseError{o.str()};
                        }
                        if (!vmats.empty()) {
                            // materials are indexed by vertex; a short list is an error reported below
                            if ((size_t)a >= vm.size() || (size_t)b >= vm.size() || (size_t)c >= vm.size())
                                throw ParseError{"Bad Trimesh: Wrong number of materials."};
                            uint32_t m3[3] = {vm[a], vm[b], vm[c]};
                            scene->addTriangle(verts[a], verts[b], verts[c], m3, true);
                        } else {
                            uint32_t m3[3] = {mm, mm, mm};
                            scene->addTriangle(verts[a], verts[b], verts[c], m3, false);
                        }
                    }
                    if (!vmats.empty() && vmats.size() != verts.size()) // trimesh.cu:80-84
                        throw ParseError{"Bad Trimesh: Wrong number of materials."};
                    return;
                }
                default: lx.fail("Expected: trimesh attributes");
            }
        }
    }

    void parseAmbientLight() { // Parser.cu:602-610: parsed and discarded
        lx.read(K_AMBIENT_LIGHT, "ambient_light"); lx.read(T_LBRACE, "{");
        if (lx.peek().kind != K_COLOR) lx.fail("Expected color attribute");
        vec3Expr();
        lx.read(T_RBRACE, "}");
    }

    void parsePointLight() { // Parser.cu:612-660
        V4 position, color;
        bool hasPos = false, hasCol = false;
        lx.read(K_POINT_LIGHT, "point_light"); lx.read(T_LBRACE, "{");
        for (;;) {
            switch (lx.peek().kind) {
                case K_POSITION:
                    if (hasPos) lx.fail("Repeated 'position' attribute");
                    position = vec3Expr(); hasPos = true; break;
                case K_COLOR:
                    if (hasCol) lx.fail("Repeated 'color' attribute");
                    color = vec3Expr(); hasCol = true; break;
                case K_CONST_ATT: case K_LIN_ATT: case K_QUAD_ATT: scalarExpr(); break;
                case T_RBRACE: {
                    if (!hasCol) lx.fail("Expected: 'color'");
                    if (!hasPos) lx.fail("Expected: 'position'");
                    lx.get();
                    // :650-651 — a light is an emissive, fully transmissive unit sphere
                    Material m;
                    m.ke = color; m.kd = V4(0, 0, 0); m.kr = V4(0, 0, 0); m.kt = V4(1, 1, 1); m.index = V4(1, 1, 1);
                    scene->addSphere(position, 1, scene->addMaterial(m));
                    return;
                }
                default:
                    lx.fail("expecting 'position' or 'color' attribute, or 'constant_attenuation_coeff', "
                            "'linear_attenuation_coeff', or 'quadratic_attenuation_coeff'");
            }
        }
    }

    void parseDirectionalLight() { // Parser.cu:662-680: parsed and discarded
        lx.read(K_DIR_LIGHT, "directional_light"); lx.read(T_LBRACE, "{");
        for (;;) {
            Tok k = lx.peek().kind;
            if (k == K_DIRECTION || k == K_COLOR) vec3Expr();
            else if (k == T_RBRACE) { lx.get(); return; }
            else lx.fail("expecting 'position' or 'color' attribute");
        }
    }

    Material parseMaterialExpression(const Material &parent) { // Parser.cu:717-723
        lx.read(K_MATERIAL, "material"); lx.read(T_EQUALS, "=");
        Material m = parseMaterial(parent);
        lx.condRead(T_SEMI);
        return m;
    }

    V4 vec3MaterialParam() { // Parser.cu:864-880
        lx.get(); lx.read(T_EQUALS, "=");
        if (lx.condRead(K_MAP)) {
            lx.read(T_LPAREN, "("); lx.read(T_IDENT, "identifier"); lx.read(T_RPAREN, ")"); lx.condRead(T_SEMI);
            return V4(1, 0, 1, 0); // texture maps are replaced by a constant (:874)
        }
        V4 v = vec3();
        lx.condRead(T_SEMI);
        return v;
    }
    float scalarMaterialParam() { // Parser.cu:882-896
        lx.get(); lx.read(T_EQUALS, "=");
        if (lx.condRead(K_MAP)) {
            lx.read(T_LPAREN, "("); lx.read(T_IDENT, "identifier"); lx.read(T_RPAREN, ")"); lx.condRead(T_SEMI);
            return 1.0f;
        }
        float v = scalar();
        lx.condRead(T_SEMI);
        return v;
    }

    Material parseMaterial(const Material &parent) { // Parser.cu:801-862
        if (lx.peek().kind == T_IDENT) // :803-805 — the identifier is looked up but NOT consumed
            return named[lx.peek().ident];
        lx.read(T_LBRACE, "{");
        std::string name;
        Material mat = parent;
        for (;;) {
            switch (lx.peek().kind) {
                case K_EMISSIVE: mat.ke = vec3MaterialParam(); break;
                case K_AMBIENT: vec3MaterialParam(); break;
                case K_SPECULAR: vec3MaterialParam(); break;
                case K_DIFFUSE: mat.kd = vec3MaterialParam(); break;
                case K_REFLECTIVE: mat.kr = vec3MaterialParam(); break;
                case K_TRANSMISSIVE: mat.kt = vec3MaterialParam(); break;
                case K_INDEX: mat.index = V4(scalarMaterialParam()); break; // setIndex(f4(float)) broadcasts, material.h:126
                case K_SHININESS: scalarMaterialParam(); break;
                case K_NAME:
                    lx.get();
                    name = lx.read(T_IDENT, "identifier").ident;
                    lx.read(T_SEMI, ";");
                    break;
                case T_RBRACE:
                    lx.get();
                    if (!name.empty()) {
                        if (named.find(name) == named.end()) named[name] = mat;
                        else lx.fail("Redefinition of material '" + name + "'.");
                    }
                    return mat;
                default: lx.fail("Expected: material attribute");
            }
        }
    }

    Lexer lx;
    std::string basePath;
    Scene *scene = nullptr;
    std::map<std::string, Material> named;
};

} // namespace

Scene *parseRayText(const std::string &text, const std::string &basePath) {
    Parser p(text, basePath);
    return p.parseScene();
}

Scene *parseRayFile(const std::string &path) {
    std::ifstream ifs(path);
    if (!ifs) throw ParseError{"Error: couldn't read scene file " + path}; // tracer.cu:44-48
    std::stringstream ss;
    ss << ifs.rdbuf();
    std::string base = path; // tracer.cu:52-56
    size_t cut = base.find_last_of("\\/");
    base = cut == std::string::npos ? "." : base.substr(0, cut);
    return parseRayText(ss.str(), base);
}

} // namespace pt
