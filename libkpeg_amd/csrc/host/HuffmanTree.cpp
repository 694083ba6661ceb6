// Host-side Huffman code tree (reference src/HuffmanTree.cpp).  The tree is grown level by
// level: at depth d the unassigned nodes, left to right, first receive the symbols of code
// length d, and every node still unassigned gets two children.  That is the canonical JPEG
// code assignment (Annex C), which the device LUTs reproduce arithmetically.
#include "HuffmanTree.hpp"

#include <vector>

#include "Logger.hpp"
#include "Utility.hpp"

namespace kpeg
{
    void insertLeft( NodePtr node, const UInt16 value )
    {
        if ( node == nullptr || node->lChild != nullptr )
            return;
        NodePtr n = createNode();
        n->parent = node;
        n->code = node->code + "0";
        n->value = value;
        node->lChild = n;
    }

    void insertRight( NodePtr node, const UInt16 value )
    {
        if ( node == nullptr || node->rChild != nullptr )
            return;
        NodePtr n = createNode();
        n->parent = node;
        n->code = node->code + "1";
        n->value = value;
        node->rChild = n;
    }

    // next node to the right on the same level, or nullptr
    NodePtr getRightLevelNode( NodePtr node )
    {
        if ( node == nullptr )
            return nullptr;
        int up = 0;
        NodePtr n = node;
        while ( n->parent != nullptr && n->parent->rChild == n )
        {
            n = n->parent;
            ++up;
        }
        if ( n->parent == nullptr )
            return nullptr;
        n = n->parent->rChild;
        while ( up-- > 0 && n != nullptr )
            n = n->lChild;
        return n;
    }

    void inOrder( NodePtr node )
    {
        if ( node == nullptr )
            return;
        inOrder( node->lChild );
        if ( node->code != "" && node->leaf )
        {
            LOG(Logger::Level::DEBUG) << "Symbol: 0x" << std::hex << node->value << ", Code: " << node->code << std::dec << std::endl;
        }
        inOrder( node->rChild );
    }

    HuffmanTree::HuffmanTree() : root_{ nullptr } {}

    HuffmanTree::HuffmanTree( const HuffmanTable& htable ) { constructHuffmanTree( htable ); }

    void HuffmanTree::constructHuffmanTree( const HuffmanTable& htable )
    {
        root_ = createRootNode( 0x0000 );
        std::vector<NodePtr> open{ root_ };  // unassigned nodes of the current depth, left to right
        for ( int len = 1; len <= 16; ++len )
        {
            std::vector<NodePtr> next;
            next.reserve( open.size() * 2 );
            for ( auto& n : open )
            {
                insertLeft( n, 0x0000 );
                insertRight( n, 0x0000 );
                next.push_back( n->lChild );
                next.push_back( n->rChild );
            }
            std::size_t used = 0;
            for ( auto&& sym : htable[len - 1].second )
            {
                if ( used >= next.size() )
                    break;  // over-subscribed table: the reference walks off the level here
                next[used]->value = sym;
                next[used]->leaf = true;
                ++used;
            }
            open.assign( next.begin() + used, next.end() );
        }
    }

    const NodePtr HuffmanTree::getTree() const { return root_; }

    const std::string HuffmanTree::contains( const std::string& huffCode )
    {
        if ( isStringWhiteSpace( huffCode ) )
        {
            LOG(Logger::Level::ERROR) << "[ FATAL ] Invalid huffman code, possibly corrupt JFIF data stream!" << std::endl;
            return "";
        }
        NodePtr n = root_;
        for ( std::size_t i = 0; i < huffCode.size() && n != nullptr; ++i )
        {
            n = huffCode[i] == '0' ? n->lChild : n->rChild;
            if ( n != nullptr && n->leaf && n->code == huffCode )
                return n->value == 0x0000 ? std::string( "EOB" ) : std::to_string( n->value );
        }
        return "";
    }
}
